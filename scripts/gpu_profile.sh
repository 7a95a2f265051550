#!/bin/bash
# Refreshes the profile evidence on the GPU box: kernel trace + stats of the headline frame (one batch in flight, like
# the profiling render bench.py takes its kernel time from), then one rocprofv3 --pmc pass per counter group over the
# headline frame and over the configs[1] kernels.  Outputs under gpurun_out/prof_<tag>/; scripts/collect_profiles.py
# condenses them into profiles/.
set -o pipefail
tag=${1:-r02}
out=$PWD/gpurun_out/prof_$tag
rm -rf "$out"
mkdir -p "$out"
export TMPDIR=/tmp
export RTGPU_RENDER_LANES=1
cd "$(dirname "$0")/.." || exit 1
HEAD="python3 bench.py --no-cpu-baseline --no-extras --no-roofline"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/trace" -- $HEAD --steps 3 --warmup 1 > "$out/bench_trace.log" 2>&1 || exit 1
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/trace_c2" -- python3 scripts/prof_c2.py > "$out/c2_trace.log" 2>&1 || exit 1
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/trace_c5" -- python3 scripts/prof_c5.py > "$out/c5_trace.log" 2>&1 || exit 1
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/trace_i1080" -- python3 scripts/prof_interior.py 1080p > "$out/i1080_trace.log" 2>&1 || exit 1
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/trace_i4k" -- python3 scripts/prof_interior.py 4k > "$out/i4k_trace.log" 2>&1 || exit 1
# one pass per group (a pass holds at most 8 SQ, 4 TCC -- FETCH_SIZE takes 3, WRITE_SIZE 2 --, 4 TCP, 2 TA counters)
groups=("FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum"
        "SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_ANY"
        "SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR"
        "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum TA_TA_BUSY_sum")
for group in "${groups[@]}"; do
    name=$(echo "$group" | cut -d' ' -f1)
    timeout -s KILL 240 rocprofv3 --pmc $group --output-format csv -d "$out/pmc_c5_$name" -- python3 scripts/prof_c5.py > "$out/c5_$name.log" 2>&1 || exit 1
    echo "configs[4] $name done"
done
for group in "${groups[@]}"; do
    name=$(echo "$group" | cut -d' ' -f1)
    timeout -s KILL 240 rocprofv3 --pmc $group --output-format csv -d "$out/pmc_head_$name" -- $HEAD --steps 1 --warmup 0 > "$out/bench_$name.log" 2>&1 || exit 1
    timeout -s KILL 200 rocprofv3 --pmc $group --output-format csv -d "$out/pmc_c2_$name" -- python3 scripts/prof_c2.py > "$out/c2_$name.log" 2>&1 || exit 1
    timeout -s KILL 200 rocprofv3 --pmc $group --output-format csv -d "$out/pmc_i1080_$name" -- python3 scripts/prof_interior.py 1080p > "$out/i1080_$name.log" 2>&1 || exit 1
    timeout -s KILL 200 rocprofv3 --pmc $group --output-format csv -d "$out/pmc_i4k_$name" -- python3 scripts/prof_interior.py 4k > "$out/i4k_$name.log" 2>&1 || exit 1
    echo "headline + configs[1] + configs[2] + configs[3] $name done"
done
echo done
