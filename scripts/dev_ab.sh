#!/bin/bash
# A/B of two builds of the library on the developer probe: scripts/dev_ab.sh <lib A> <lib B> [args of dev_kernel_perf.py]
a=$1; b=$2; shift 2
for lib in "$a" "$b" "$a" "$b"; do
  RTGPU_LIB=$PWD/$lib timeout -k 10 200 python scripts/dev_kernel_perf.py "$@" 2>&1 | tail -1 || exit 1
done
