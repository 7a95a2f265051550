#!/bin/bash
for m in 1 4 8 16 32 48; do
  echo "minIdle=$m"
  RTGPU_STEAL_MIN_IDLE=$m timeout -k 10 120 python scripts/dev_perf.py --what trace 2>&1 | grep -E "primary|bounce|random inc" || exit 1
done
echo "no stealing"
RTGPU_TRACE_VARIANT=-1 timeout -k 10 120 python scripts/dev_perf.py --what trace 2>&1 | grep -E "primary|bounce|random inc"
