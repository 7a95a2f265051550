#!/bin/bash
# L1 (TCP) / texture-addresser counters of the configs[1] kernels: is the vector memory path of a CU what the traversal waits for?
set -o pipefail
out=$PWD/gpurun_out/pmc_l1
mkdir -p "$out"
export TMPDIR=/tmp
cd "$(dirname "$0")/.." || exit 1
for group in "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum" "TA_TA_BUSY_sum TA_BUSY_avr" "TD_TD_BUSY_sum GRBM_GUI_ACTIVE" "TCP_TOTAL_ACCESSES_sum TCP_TA_TCP_STATE_READ_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum"; do
    name=$(echo "$group" | cut -d' ' -f1)
    timeout -s KILL 200 rocprofv3 --pmc $group --output-format csv -d "$out/$name" -- python3 scripts/prof_c2.py > "$out/$name.log" 2>&1 || { tail -5 "$out/$name.log"; exit 1; }
done
python3 - <<'PY'
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("gpurun_out/pmc_l1/*/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "steal" in r["Kernel_Name"]:
            acc[r["Kernel_Name"][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in acc.items():
    print(k)
    for c, vals in sorted(v.items()):
        n = len(vals)
        print("   %-45s n=%d first half mean %.4g second half mean %.4g" % (c, n, sum(vals[:n//2]) / max(n//2, 1), sum(vals[n//2:]) / max(n - n//2, 1)))
PY
