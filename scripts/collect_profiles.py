#!/usr/bin/env python3
"""Condenses gpurun_out/prof_<tag>/ (written by scripts/gpu_profile.sh) into tracked files under
profiles/: the kernel-stats CSV, a per-launch PMC summary of the dominant kernel and the HBM
traffic figure bench.py reports as roofline.traffic.

HBM bytes per launch = 2 x FETCH_SIZE + WRITE_SIZE (both in KiB): on gfx950 FETCH_SIZE tallies
128-B requests at 64 B, WRITE_SIZE is exact.  The x2 is calibrated for wide streaming reads; for
this kernel's 16-B gathers it is an upper bound.
"""
import collections
import csv
import glob
import json
import os
import shutil
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
KERNEL = "trace_closest_steal_kernel<false>"


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
    src = os.path.join(ROOT, "gpurun_out", "prof_" + tag)
    dst = os.path.join(ROOT, "profiles")
    stats = glob.glob(os.path.join(src, "trace", "*", "*_kernel_stats.csv"))[0]
    shutil.copy(stats, os.path.join(dst, tag + "_bench_kernel_stats.csv"))
    trace = glob.glob(os.path.join(src, "trace", "*", "*_kernel_trace.csv"))[0]
    with open(trace) as f, open(os.path.join(dst, tag + "_bench_kernel_trace_trace_closest.csv"), "w") as g:
        rd = csv.DictReader(f)
        wr = csv.writer(g)
        wr.writerow(["Kernel_Name", "Start_Timestamp", "End_Timestamp", "DurationNs", "Grid_Size", "Workgroup_Size", "LDS_Block_Size", "VGPR_Count", "SGPR_Count"])
        for r in rd:
            if "rt::" in r["Kernel_Name"]:
                wr.writerow([r["Kernel_Name"].split("(")[0], r["Start_Timestamp"], r["End_Timestamp"],
                             int(r["End_Timestamp"]) - int(r["Start_Timestamp"]), r.get("Grid_Size", r.get("Grid_Size_X", "")),
                             r.get("Workgroup_Size", r.get("Workgroup_Size_X", "")), r.get("LDS_Block_Size", ""),
                             r.get("VGPR_Count", ""), r.get("SGPR_Count", "")])
    summary = {"kernel": KERNEL, "per_launch": {}}
    avg_ns = None
    for r in csv.DictReader(open(stats)):
        if KERNEL in r["Name"]:
            avg_ns = float(r["AverageNs"])
            summary["kernel_trace"] = {"calls": int(r["Calls"]), "average_ns": avg_ns, "min_ns": int(r["MinNs"]), "max_ns": int(r["MaxNs"])}
    for d in sorted(glob.glob(os.path.join(src, "pmc_*"))):
        files = glob.glob(os.path.join(d, "*", "*_counter_collection.csv"))
        if not files:
            continue
        acc = collections.defaultdict(list)
        for r in csv.DictReader(open(files[0])):
            if KERNEL in r["Kernel_Name"]:
                acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, v in acc.items():
            summary["per_launch"][k] = {"mean": sum(v) / len(v), "min": min(v), "max": max(v), "launches": len(v)}
    pl = summary["per_launch"]
    if "FETCH_SIZE" in pl and "WRITE_SIZE" in pl:
        hbm = (2.0 * pl["FETCH_SIZE"]["mean"] + pl["WRITE_SIZE"]["mean"]) * 1024.0
        summary["hbm_bytes_per_launch"] = hbm
        json.dump({"kernel": KERNEL, "hbm_bytes_per_launch": hbm,
                   "fetch_size_kib": pl["FETCH_SIZE"]["mean"], "write_size_kib": pl["WRITE_SIZE"]["mean"],
                   "formula": "(2*FETCH_SIZE + WRITE_SIZE) * 1024; FETCH_SIZE doubled per the gfx950 correction (upper bound for 16-B gathers)",
                   "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE, separate runs of bench.py --steps 5 --warmup 1 --no-cpu-baseline (%s)" % tag},
                  open(os.path.join(dst, "traffic_trace_closest.json"), "w"), indent=1)
    json.dump(summary, open(os.path.join(dst, tag + "_pmc_summary.json"), "w"), indent=1)
    print(json.dumps(summary, indent=1))


if __name__ == "__main__":
    main()
