#!/bin/bash
# developer comparison of library builds (raytracer_amd/librtgpu_<name>.so) on the kernel perf probe
timeout -k 10 150 python scripts/dev_kernel_perf.py || exit 1
for v in "$@"; do
  RTGPU_LIB=$PWD/raytracer_amd/librtgpu_$v.so timeout -k 10 150 python scripts/dev_kernel_perf.py || exit 1
done
