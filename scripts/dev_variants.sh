#!/bin/bash
# developer probe: launch-shape variants of the closest-hit kernel on the C2 workload
for v in 0 1 2 3; do
  for w in 2560 5120 10240; do
    if [ $v -lt 2 ] && [ $w -ne 5120 ]; then continue; fi
    ww=$w; if [ $v -eq 3 ]; then ww=$((w/4)); fi
    echo "variant=$v waves=$ww"
    RTGPU_TRACE_VARIANT=$v RTGPU_TRACE_WAVES=$ww timeout -k 10 120 python scripts/dev_perf.py --what trace 2>&1 | grep -E "primary|bounce|random inc" || exit 1
  done
done
