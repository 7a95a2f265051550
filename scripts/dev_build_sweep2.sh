#!/bin/bash
# sweep of mesh-tree build parameters against the cooperative kernels (configs[1] rays, no frame)
for leaf in 2 4 6 8; do for tc in 0.5 1 2; do
  echo -n "maxLeaf=$leaf traversalCost=$tc  "
  RTGPU_BUILD_MAX_LEAF=$leaf RTGPU_BUILD_TRAVERSAL_COST=$tc timeout -k 10 120 python scripts/dev_kernel_perf.py --no-frame 2>&1 | tail -1 || exit 1
done; done
