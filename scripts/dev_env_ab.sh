#!/bin/bash
# A/B of an environment switch on the developer probes: scripts/dev_env_ab.sh VAR=value [VAR=value ...]
for k in 1 2; do
  timeout -k 10 200 python scripts/dev_kernel_perf.py 2>&1 | tail -1 | cut -c1-300 || exit 1
  env "$@" timeout -k 10 200 python scripts/dev_kernel_perf.py 2>&1 | tail -1 | cut -c1-300 || exit 1
done
timeout -k 10 200 python scripts/dev_c5_frame.py 2>&1 | tail -1 || exit 1
env "$@" timeout -k 10 200 python scripts/dev_c5_frame.py 2>&1 | tail -1 || exit 1
