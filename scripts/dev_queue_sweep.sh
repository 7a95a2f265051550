#!/bin/bash
# sweep of the persistent grid size and fetch threshold of the ray-queue traversal
echo "non-persistent stealing kernel"
RTGPU_TRACE_VARIANT=4 timeout -k 10 120 python scripts/dev_perf.py --what trace 2>&1 | grep -E "primary|bounce|random" || exit 1
for g in ${GRIDS:-2048 4096 6144 7168 8192 12288 16384}; do for f in ${FETCH:-8}; do
  echo "waves=$g fetchMin=$f"
  RTGPU_QUEUE_WAVES=$g RTGPU_QUEUE_FETCH_MIN=$f timeout -k 10 120 python scripts/dev_perf.py --what trace 2>&1 | grep -E "primary|bounce|random" || exit 1
done; done
