#!/bin/bash
# sweep of the work-stealing parameters on the C3 frame (incoherent extension and shadow rays)
for cfg in "1 48" "1 32" "1 64" "1 80" "1 100" "4 48" "8 48" "16 48"; do
  set -- $cfg
  echo "minIdle=$1 leafPct=$2"
  RTGPU_STEAL_MIN_IDLE=$1 RTGPU_STEAL_LEAF_PCT=$2 timeout -k 10 120 python scripts/dev_perf.py --what render 2>&1 | grep -E "render" || exit 1
done
