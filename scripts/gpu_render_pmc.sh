#!/bin/bash
# PMC passes over one C3-style render; prints per-kernel totals
export TMPDIR=/tmp
for pass in "SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "FETCH_SIZE" "WRITE_SIZE"; do
  tag=$(echo $pass | cut -d' ' -f1)
  out=$PWD/gpurun_out/pmc_render_$tag
  rm -rf "$out"
  timeout -s KILL 200 rocprofv3 --pmc $pass --output-format csv -d "$out" -- python3 scripts/dev_perf.py --what render > "$out.log" 2>&1 || exit 1
  python3 - "$out" <<'PY'
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + "/*/*_counter_collection.csv")[0]
acc = collections.defaultdict(lambda: collections.defaultdict(float))
n = collections.defaultdict(int)
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"]
    if "rt::" not in k: continue
    k = k.split("_kernel")[0].split("::")[-1]
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
for k, v in acc.items():
    print(k, " ".join("%s=%.4g" % (c, x) for c, x in sorted(v.items())))
PY
done
