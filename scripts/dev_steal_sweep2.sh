#!/bin/bash
# stealing threshold / leaf-step threshold against the quantised-node kernels
for mi in 1 4 8 16; do echo -n "min idle $mi: "; RTGPU_STEAL_MIN_IDLE=$mi timeout -k 10 120 python scripts/dev_kernel_perf.py 2>&1 | tail -1 | cut -c1-200 || exit 1; done
for lp in 24 32 64 80; do echo -n "leaf pct $lp: "; RTGPU_STEAL_LEAF_PCT=$lp timeout -k 10 120 python scripts/dev_kernel_perf.py 2>&1 | tail -1 | cut -c1-200 || exit 1; done
