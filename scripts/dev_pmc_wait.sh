#!/bin/bash
# Developer probe: memory-instruction counts and accumulated latencies of the shade kernels on the 1080p frame.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export RTGPU_RENDER_LANES=1
timeout -s KILL 200 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_LEVEL_VMEM SQ_INSTS_SMEM SQ_INST_LEVEL_SMEM SQ_WAVE_CYCLES SQ_WAIT_INST_ANY --output-format csv -d gpurun_out/wait1 -- python3 scripts/dev_kernel_perf.py > gpurun_out/wait1.log 2>&1 || exit 1
timeout -s KILL 200 rocprofv3 --pmc TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_ATOMIC_WITH_RET_REQ_sum --output-format csv -d gpurun_out/wait2 -- python3 scripts/dev_kernel_perf.py > gpurun_out/wait2.log 2>&1 || exit 1
timeout -s KILL 200 rocprofv3 --pmc SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SMEM SQ_IFETCH SQ_IFETCH_LEVEL SQ_BUSY_CYCLES --output-format csv -d gpurun_out/wait3 -- python3 scripts/dev_kernel_perf.py > gpurun_out/wait3.log 2>&1 || exit 1
