#!/bin/bash
# developer sweep of the stealing parameters on the kernel perf probe
for mi in 1 4 8 16 24; do
  echo "minIdle=$mi"; RTGPU_STEAL_MIN_IDLE=$mi timeout -k 10 120 python scripts/dev_kernel_perf.py || exit 1
done
for lp in 24 32 64 96; do
  echo "leafPct=$lp"; RTGPU_STEAL_LEAF_PCT=$lp timeout -k 10 120 python scripts/dev_kernel_perf.py || exit 1
done
