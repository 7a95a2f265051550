#!/bin/bash
# What the shade kernels wait for: instruction fetch, scalar / vector memory, LDS, export -- PMC passes over one 1080p frame.
export TMPDIR=/tmp
cd "$(dirname "$0")/.." || exit 1
for pass in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_IFETCH SQ_WAIT_IFETCH SQ_INSTS_VALU" \
            "SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_BRANCH SQ_INSTS_SALU SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_SMEM SQ_INST_LEVEL_LDS" \
            "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU" \
            "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES"; do
  tag=$(echo $pass | cut -d' ' -f1)
  out=$PWD/gpurun_out/pmc_shade_$tag
  rm -rf "$out"
  timeout -s KILL 120 rocprofv3 --pmc $pass --output-format csv -d "$out" -- python3 scripts/dev_kernel_perf.py --no-trace > "$out.log" 2>&1 || { echo "pass $tag failed"; tail -3 "$out.log"; continue; }
  python3 - "$out" <<'PY'
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + "/*/*_counter_collection.csv")[0]
acc = collections.defaultdict(lambda: collections.defaultdict(float))
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"]
    if "rt::" not in k: continue
    k = ("shade_first" if "shade_kernel<true" in k else "shade_later" if "shade_kernel<false" in k else k.split("_kernel")[0].split("::")[-1])
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
for k, v in sorted(acc.items()):
    print(k, " ".join("%s=%.4g" % (c, x) for c, x in sorted(v.items())))
PY
done
