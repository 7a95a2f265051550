#!/bin/bash
# Refreshes the profile evidence of the bench workload on the GPU box: kernel trace + stats, then
# one PMC pass per counter group.  Outputs under gpurun_out/prof_<tag>/.
set -o pipefail
tag=${1:-r01}
out=$PWD/gpurun_out/prof_$tag
mkdir -p "$out"
export TMPDIR=/tmp
cd "$(dirname "$0")/.." || exit 1
timeout -k 10 240 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/trace" -- python3 bench.py --steps 20 --no-cpu-baseline --no-frame > "$out/bench_trace.log" 2>&1 || exit 1
timeout -s KILL 180 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$out/pmc_fetch" -- python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-frame > "$out/bench_fetch.log" 2>&1 || exit 1
timeout -s KILL 180 rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$out/pmc_write" -- python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-frame > "$out/bench_write.log" 2>&1 || exit 1
timeout -s KILL 180 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d "$out/pmc_l2" -- python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-frame > "$out/bench_l2.log" 2>&1 || exit 1
timeout -s KILL 180 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_BUSY_CYCLES SQ_INSTS_LDS SQ_WAIT_INST_ANY --output-format csv -d "$out/pmc_sq" -- python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-frame > "$out/bench_sq.log" 2>&1 || exit 1
echo done
