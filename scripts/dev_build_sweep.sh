#!/bin/bash
# sweep of mesh-tree build parameters against traversal speed
for leaf in 2 3 4 6 8; do for tc in 0.5 1 2; do
  echo "maxLeaf=$leaf traversalCost=$tc"
  RTGPU_BUILD_MAX_LEAF=$leaf RTGPU_BUILD_TRAVERSAL_COST=$tc timeout -k 10 120 python scripts/dev_perf.py --what trace 2>&1 | grep -E "blob|primary|bounce|random inc" || exit 1
done; done
