#!/usr/bin/env python3
"""Per-kernel instruction mix from gpurun_out/mix_<tag>/ (scripts/gpu_pmc_mix.sh): counter sums over one headline frame."""
import collections, csv, glob, os, re, sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r02"
src = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", "mix_" + tag)
names = ["primary_kernel", "shade_kernel", "trace_stage_kernel", "resolve_kernel"]


def key(n):
    for k in names:
        if k in n:
            m = re.search(k + r"<([^>]*)>", n)
            return k + ("<" + m.group(1) + ">" if m else "")
    return None


acc = collections.defaultdict(lambda: collections.defaultdict(float))
for f in glob.glob(os.path.join(src, "g*", "*", "*_counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        k = key(r["Kernel_Name"])
        if k:
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
for k, c in sorted(acc.items()):
    v = max(1.0, c.get("SQ_INSTS_VALU", 0))
    print(k)
    print("   waves %.3g, VALU per wave %.0f, active lanes per VALU instruction %.1f" % (c["SQ_WAVES"], v / max(1, c["SQ_WAVES"]), c["SQ_THREAD_CYCLES_VALU"] / max(1, c["SQ_ACTIVE_INST_VALU"])))
    print("   VALU mix: " + ", ".join("%s %.1f%%" % (n[14:], 100 * c.get(n, 0) / v) for n in sorted(c) if n.startswith("SQ_INSTS_VALU_")))
    print("   per VALU: SALU %.2f, LDS %.3f, VMEM rd %.3f wr %.3f, SMEM %.3f, branch %.3f, ifetch %.3f" % tuple(c.get(n, 0) / v for n in ("SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR", "SQ_INSTS_SMEM", "SQ_INSTS_BRANCH", "SQ_IFETCH")))
    print("   cycles: VALU active %.3g, scalar active %.3g, LDS active %.3g (bank conflict %.3g), VMEM active %.3g, busy %.3g, wave cycles %.3g, wait any %.3g, wait inst %.3g, wait LDS %.3g"
          % tuple(c.get(n, 0) for n in ("SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_SCA", "SQ_ACTIVE_INST_LDS", "SQ_LDS_BANK_CONFLICT", "SQ_ACTIVE_INST_VMEM", "SQ_BUSY_CYCLES", "SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_WAIT_INST_LDS")))
