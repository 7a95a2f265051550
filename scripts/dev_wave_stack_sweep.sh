#!/bin/bash
# Developer sweep: LDS stack size of the one-wave-per-query gather (library variants built with -DRT_WAVE_STACK=n) x first-attempt budget
for w in ${STACKS:-192 320 448}; do for b in ${BUDGETS:-128 192 256 384}; do for d in 0 1; do
  echo -n "stack $w budget $b: "
  RTGPU_LIB=$PWD/raytracer_amd/librtgpu_ws$w.so RTGPU_GATHER_BUDGET=$b timeout -k 10 120 python -u scripts/dev_photon_frame.py $d 2>&1 | grep photons | sed 's/photons [0-9]*, render 1080p 4x4 spp //' || exit 1
done; done; done
