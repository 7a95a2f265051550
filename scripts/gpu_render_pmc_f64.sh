#!/bin/bash
# One PMC pass over a C3-style render: VALU instructions by type (fp64 share of the shade kernel)
export TMPDIR=/tmp
out=$PWD/gpurun_out/pmc_render_f64
rm -rf "$out"
timeout -s KILL 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 --output-format csv -d "$out" -- python3 scripts/dev_perf.py --what render > "$out.log" 2>&1 || { tail -5 "$out.log"; exit 1; }
python3 - "$out" <<'PY'
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + "/*/*_counter_collection.csv")[0]
acc = collections.defaultdict(lambda: collections.defaultdict(float))
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"]
    if "rt::" not in k: continue
    k = k.split("_kernel")[0].split("::")[-1]
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
for k, v in acc.items():
    print(k, " ".join("%s=%.4g" % (c.replace("SQ_INSTS_VALU", "V"), x) for c, x in sorted(v.items())))
PY
