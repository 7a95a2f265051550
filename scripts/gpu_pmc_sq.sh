#!/bin/bash
# one SQ counter pass over bench.py; prints per-launch means of every rt:: trace kernel
out=$PWD/gpurun_out/pmc_sq_${1:-x}
export TMPDIR=/tmp
timeout -s KILL 180 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_ANY --output-format csv -d "$out" -- python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-frame > "$out.log" 2>&1 || exit 1
python3 - "$out" <<'PY'
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + "/*/*_counter_collection.csv")[0]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f)):
    if "rt::trace" in r["Kernel_Name"]:
        acc[r["Kernel_Name"].split("(")[0]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in acc.items():
    print(k)
    for c, x in sorted(v.items()):
        print("   %-22s %14.0f (%d launches)" % (c, sum(x) / len(x), len(x)))
PY
