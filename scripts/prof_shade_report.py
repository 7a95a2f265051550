#!/usr/bin/env python3
"""Prints, per kernel of gpurun_out/prof_<tag>/ (scripts/prof_shade.sh), time and counter-derived shares."""
import collections, csv, glob, os, sys
tag = sys.argv[1] if len(sys.argv) > 1 else "shade"
root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", "prof_" + tag)
times = {}
for f in glob.glob(os.path.join(root, "trace", "*", "*kernel_stats.csv")):
    for r in csv.DictReader(open(f)):
        times[r["Name"]] = (int(r["Calls"]), float(r["TotalDurationNs"]) * 1e-9)
cnt = collections.defaultdict(lambda: collections.defaultdict(float))
for f in glob.glob(os.path.join(root, "pmc_*", "*", "*_counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        cnt[r["Kernel_Name"]][r["Counter_Name"]] += float(r["Counter_Value"])
def short(n):
    return n.replace("rt::(anonymous namespace)::", "").split("(")[0][:40]
for name, (calls, sec) in sorted(times.items(), key=lambda kv: -kv[1][1]):
    c = cnt.get(name)
    if not c or sec < 1e-4:
        continue
    cyc = sec * 2.4e9
    line = "%-40s %4d launches %8.2f ms" % (short(name), calls, sec * 1e3)
    if "SQ_INSTS_VALU" in c:
        line += " | VALU issue %.2f  inst/wave %.0f  waves %.2f M  wait %.2f  wait_inst %.2f  resident %.1f/SIMD" % (
            c["SQ_INSTS_VALU"] * 4 / 1024 / cyc, c["SQ_INSTS_VALU"] / max(1, c["SQ_WAVES"]), c["SQ_WAVES"] / 1e6,
            c["SQ_WAIT_ANY"] / max(1, c["SQ_WAVE_CYCLES"]), c["SQ_WAIT_INST_ANY"] / max(1, c["SQ_WAVE_CYCLES"]), c["SQ_WAVE_CYCLES"] / 1024 / cyc * 4 if False else c["SQ_WAVE_CYCLES"] / (1024 * cyc))
        line += "  salu/valu %.2f lds/valu %.3f" % (c["SQ_INSTS_SALU"] / c["SQ_INSTS_VALU"], c["SQ_INSTS_LDS"] / c["SQ_INSTS_VALU"])
    if "SQ_ACTIVE_INST_VALU" in c and c["SQ_ACTIVE_INST_VALU"]:
        line += " | lanes %.1f  vmem rd %.1f wr %.1f /wave" % (c["SQ_THREAD_CYCLES_VALU"] / c["SQ_ACTIVE_INST_VALU"],
                                                             c["SQ_INSTS_VMEM_RD"] / max(1, c["SQ_WAVES"]), c["SQ_INSTS_VMEM_WR"] / max(1, c["SQ_WAVES"]))
    if "TCP_TOTAL_CACHE_ACCESSES_sum" in c:
        line += " | L1 %.1f TB/s hit %.2f  TA busy %.2f  pending stall %.2f" % (c["TCP_TOTAL_CACHE_ACCESSES_sum"] * 64 / sec / 1e12,
            1 - c["TCP_TCC_READ_REQ_sum"] / max(1, c["TCP_TOTAL_CACHE_ACCESSES_sum"]), c["TA_TA_BUSY_sum"] / 256 / cyc, c["TCP_PENDING_STALL_CYCLES_sum"] / 256 / cyc)
    if "FETCH_SIZE" in c:
        line += " | HBM side <= %.2f TB/s (fetch %.1f GB, write %.1f GB)" % ((2 * c["FETCH_SIZE"] + c.get("WRITE_SIZE", 0)) * 1024 / sec / 1e12,
                                                                          c["FETCH_SIZE"] * 1024 / 1e9, c.get("WRITE_SIZE", 0) * 1024 / 1e9)
    print(line)
