#!/bin/bash
# sweep of the resident paths per batch and of the batches in flight on the C3 frame
for cfg in "8388608 2" "4194304 2" "16777216 2" "33554432 2" "67108864 2" "16777216 3" "33554432 3" "8388608 3" "8388608 4"; do
  set -- $cfg
  echo "paths=$1 lanes=$2"
  RTGPU_BATCH_PATHS=$1 RTGPU_RENDER_LANES=$2 timeout -k 10 120 python scripts/dev_perf.py --what render 2>&1 | grep -E "render" | tail -1 || exit 1
done
