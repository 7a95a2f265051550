#!/bin/bash
# configs[4] frame against batch size / batches in flight
timeout -k 10 200 python scripts/dev_c5_frame.py 2>&1 | tail -1 || exit 1
RTGPU_BATCH_PATHS=8388608 RTGPU_RENDER_LANES=4 timeout -k 10 200 python scripts/dev_c5_frame.py 2>&1 | tail -1 || exit 1
RTGPU_BATCH_PATHS=4194304 RTGPU_RENDER_LANES=8 timeout -k 10 200 python scripts/dev_c5_frame.py 2>&1 | tail -1 || exit 1
RTGPU_BATCH_PATHS=4194304 RTGPU_RENDER_LANES=4 timeout -k 10 200 python scripts/dev_c5_frame.py 2>&1 | tail -1 || exit 1
RTGPU_BATCH_PATHS=2097152 RTGPU_RENDER_LANES=8 timeout -k 10 200 python scripts/dev_c5_frame.py 2>&1 | tail -1 || exit 1
