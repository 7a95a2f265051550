#!/bin/bash
# configs[4] frame against mesh-tree build parameters and stealing thresholds
for leaf in 4 8 12; do for tc in 1 2; do
  RTGPU_BUILD_MAX_LEAF=$leaf RTGPU_BUILD_TRAVERSAL_COST=$tc timeout -k 10 200 python scripts/dev_c5_frame.py 2>&1 | tail -1 || exit 1
done; done
for idle in 1 8 32; do
  RTGPU_STEAL_MIN_IDLE=$idle timeout -k 10 200 python scripts/dev_c5_frame.py 2>&1 | tail -1 || exit 1
done
