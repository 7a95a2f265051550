#!/bin/bash
# Instruction-mix counters of the headline frame's kernels (one rocprofv3 --pmc pass per group of at most 8 SQ counters).
# Outputs under gpurun_out/mix_<tag>/; scripts/pmc_mix_report.py prints the per-kernel table.
set -o pipefail
tag=${1:-r02}
out=$PWD/gpurun_out/mix_$tag
mkdir -p "$out"
export TMPDIR=/tmp
export RTGPU_RENDER_LANES=1
cd "$(dirname "$0")/.." || exit 1
HEAD="python3 bench.py --no-cpu-baseline --no-extras --no-roofline --steps 1 --warmup 0"
i=0
for group in \
  "SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_BUSY_CYCLES SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT" \
  "SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64" \
  "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_SCA SQ_INSTS_SALU SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_INSTS_BRANCH" \
  "SQ_IFETCH SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY"; do
    i=$((i + 1))
    timeout -s KILL 240 rocprofv3 --pmc $group --output-format csv -d "$out/g$i" -- $HEAD > "$out/g$i.log" 2>&1 || exit 1
done
echo done
