#!/bin/bash
# Developer sweep: first-attempt budget of the photon gather (one photon-mapped frame each; the tail goes to whole waves).
set -e
for b in 256 512 1024 2048 4096 8192; do
  echo "== budget $b"
  RTGPU_GATHER_BUDGET=$b timeout -k 10 120 python -u scripts/dev_photon_frame.py ${1:-0} 2>&1 | grep photons
done
