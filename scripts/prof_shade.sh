#!/bin/bash
# Counters of the shade kernels on the GPU box: kernel trace, then the two SQ groups and the L1 group, over one 1080p
# frame of the headline scene (one batch in flight).  scripts/prof_shade_report.py prints the shares.
set -o pipefail
tag=${1:-shade}
out=$PWD/gpurun_out/prof_$tag
rm -rf "$out"
mkdir -p "$out"
export TMPDIR=/tmp
export RTGPU_RENDER_LANES=1
cd "$(dirname "$0")/.." || exit 1
HEAD="python3 bench.py --no-cpu-baseline --no-extras --no-roofline --width 1920 --height 1080 --steps 1 --warmup 0"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/trace" -- $HEAD > "$out/trace.log" 2>&1 || exit 1
groups=("SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_ANY"
        "SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM_RD SQ_INSTS_SMEM"
        "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum TA_TA_BUSY_sum"
        "FETCH_SIZE" "WRITE_SIZE")
for group in "${groups[@]}"; do
    name=$(echo "$group" | cut -d' ' -f1)
    timeout -s KILL 240 rocprofv3 --pmc $group --output-format csv -d "$out/pmc_$name" -- $HEAD > "$out/$name.log" 2>&1 || exit 1
    echo "$name done"
done
