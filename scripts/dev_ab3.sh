#!/bin/bash
# several builds of the library on the developer probe, twice round: scripts/dev_ab3.sh "<lib> <lib> ..." [args of dev_kernel_perf.py]
libs=$1; shift
for round in 1 2; do for lib in $libs; do
  RTGPU_LIB=$PWD/$lib timeout -k 10 200 python scripts/dev_kernel_perf.py "$@" 2>&1 | tail -1 || exit 1
done; done
