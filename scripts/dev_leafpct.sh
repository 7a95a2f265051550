#!/bin/bash
# sweep of the leaf-step share of the work-stealing traversal
for p in ${PCTS:-8 16 32 48 64}; do
  echo "leafPct=$p"
  RTGPU_STEAL_LEAF_PCT=$p timeout -k 10 120 python scripts/dev_perf.py --what trace 2>&1 | grep -E "primary|bounce|random" || exit 1
done
