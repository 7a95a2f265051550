#!/bin/bash
# Memory-side counters of the shade kernels over three 1080p frames of the headline scene (scripts/dev_kernel_perf.py
# --no-trace): fabric bytes, L2 hits / misses, L2 <-> memory requests by size, L1 and texture-addresser busy.
export TMPDIR=/tmp
cd "$(dirname "$0")/.." || exit 1
for pass in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" \
            "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum" \
            "TA_TA_BUSY_sum TA_BUSY_avr" "GRBM_GUI_ACTIVE" \
            "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_PENDING_STALL_CYCLES_sum"; do
  tag=$(echo $pass | cut -d' ' -f1)
  out=$PWD/gpurun_out/pmc_shmem_$tag
  rm -rf "$out"
  timeout -s KILL 120 rocprofv3 --pmc $pass --output-format csv -d "$out" -- python3 scripts/dev_kernel_perf.py --no-trace > "$out.log" 2>&1 || { echo "pass $tag failed"; tail -3 "$out.log"; continue; }
  python3 - "$out" <<'PY'
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + "/*/*_counter_collection.csv")[0]
acc = collections.defaultdict(lambda: collections.defaultdict(float))
n = collections.Counter()
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"]
    if "rt::" not in k: continue
    k = ("shade_first" if "shade_kernel<true" in k else "shade_later" if "shade_kernel<false" in k else k.split("_kernel")[0].split("::")[-1])
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
for k, v in sorted(acc.items()):
    print(k, " ".join("%s=%.5g" % (c, x) for c, x in sorted(v.items())))
PY
done
