#!/bin/bash
# Developer sweep: first-attempt budget and tail grid of the photon gather (one photon-mapped frame each).
set -e
for cfg in "2048 4096" "2048 65536" "8192 65536" "32768 65536" "4000000000 4096"; do
  set -- $cfg
  echo "== budget $1, tail blocks $2"
  RTGPU_GATHER_BUDGET=$1 RTGPU_GATHER_TAIL_BLOCKS=$2 RTGPU_GATHER_STATS=1 timeout -k 10 120 python -u scripts/dev_photon_frame.py 2>&1 | grep -v amdgpu.ids | sort | uniq -c | sort -rn | head -8
done
