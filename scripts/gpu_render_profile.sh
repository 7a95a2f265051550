#!/bin/bash
# kernel-trace of one C3-style render; prints total time per kernel
out=$PWD/gpurun_out/prof_render_${1:-x}
export TMPDIR=/tmp
rm -rf "$out"
timeout -k 10 240 rocprofv3 --kernel-trace --stats --output-format csv -d "$out" -- python3 scripts/dev_perf.py --what render > "$out.log" 2>&1 || exit 1
grep render "$out.log"
python3 - "$out" <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/*/*_kernel_stats.csv")[0]
rows = [r for r in csv.DictReader(open(f)) if "rt::" in r["Name"]]
tot = sum(float(r["TotalDurationNs"]) for r in rows)
for r in rows:
    print("%-40s calls %5s total %9.2f ms  avg %9.1f us  %5.1f%%" % (r["Name"].split("_kernel")[0].split("::")[-1], r["Calls"], float(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e3, 100 * float(r["TotalDurationNs"]) / tot))
print("sum of kernels %.2f ms (2 renders)" % (tot / 1e6))
PY
