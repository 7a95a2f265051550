#!/bin/bash
# PMC passes over one photon-mapped frame; prints per-kernel totals of the gather kernels
export TMPDIR=/tmp
for pass in "SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "FETCH_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
  tag=$(echo $pass | cut -d' ' -f1)
  out=$PWD/gpurun_out/pmc_gather_$tag
  rm -rf "$out"
  RTGPU_RENDER_LANES=1 timeout -s KILL 200 rocprofv3 --pmc $pass --output-format csv -d "$out" -- python3 scripts/dev_photon_frame.py > "$out.log" 2>&1 || { tail -3 "$out.log"; exit 1; }
  python3 - "$out" <<'PY'
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + "/*/*_counter_collection.csv")[0]
acc = collections.defaultdict(lambda: collections.defaultdict(float))
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"]
    if "gather" not in k: continue
    k = k.split("_kernel")[0].split("::")[-1]
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
for k, v in acc.items():
    print(k, " ".join("%s=%.4g" % (c, x) for c, x in sorted(v.items())))
PY
done
