#!/usr/bin/env python3
"""Condenses gpurun_out/prof_<tag>/ (written by scripts/gpu_profile.sh) into tracked files under profiles/:
<tag>_bench_kernel_stats.csv / <tag>_c2_kernel_stats.csv (rocprofv3 --kernel-trace --stats) and <tag>_pmc_summary.json,
the per-frame (headline trace stage) and per-launch (configs[1] kernels) counter figures bench.py quotes beside its
live numbers.

HBM bytes = 2 x FETCH_SIZE + WRITE_SIZE (both in KiB): on gfx950 FETCH_SIZE tallies 128-B requests at 64 B, WRITE_SIZE
is exact.  The x2 is calibrated for wide streaming reads; for these kernels' 16-B gathers it is an upper bound.
VALU issue share = SQ_INSTS_VALU x 4 cycles / (1024 SIMDs x kernel time x 2.4 GHz).
Lanes per VALU instruction = SQ_THREAD_CYCLES_VALU / SQ_ACTIVE_INST_VALU.  L2 -> compute-unit traffic = TCP_TCC_READ_REQ x 64 B
(the L1s' read requests to L2), L1 accesses = TCP_TOTAL_CACHE_ACCESSES (64-byte look-ups).
"""
import collections
import csv
import glob
import json
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def newest(pattern):
    """Files matching `pattern`, the most recently written first (a directory that was profiled twice holds both runs)."""
    return sorted(glob.glob(pattern), key=os.path.getmtime, reverse=True)


def counters(dirs, kernel, split=None):
    """Sums (per run) or lists (split = launches per group) the counter values of `kernel` over the runs in `dirs`."""
    out = {}
    for d in dirs:
        files = newest(os.path.join(d, "*", "*_counter_collection.csv"))
        if not files:
            continue
        acc = collections.defaultdict(list)
        for r in csv.DictReader(open(files[0])):
            if kernel in r["Kernel_Name"]:
                acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, v in acc.items():
            out[k] = v
    return out


def kernel_time(stats_csv, kernel):
    for r in csv.DictReader(open(stats_csv)):
        if kernel in r["Name"]:
            return {"calls": int(r["Calls"]), "average_ns": float(r["AverageNs"]), "total_ns": float(r["TotalDurationNs"]),
                    "min_ns": int(r["MinNs"]), "max_ns": int(r["MaxNs"])}
    return None


def derived(c, seconds):
    o = {}
    if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
        o["fetch_size_kib"], o["write_size_kib"] = c["FETCH_SIZE"], c["WRITE_SIZE"]
        o["hbm_bytes"] = (2.0 * c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024.0
        if seconds:
            o["hbm_GBps"] = o["hbm_bytes"] / seconds / 1e9
    if "TCC_HIT_sum" in c and "TCC_MISS_sum" in c:
        o["l2_hit_rate"] = c["TCC_HIT_sum"] / max(1.0, c["TCC_HIT_sum"] + c["TCC_MISS_sum"])
    for k in ("SQ_WAVES", "SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_ACTIVE_INST_ANY", "SQ_WAIT_INST_ANY",
              "SQ_THREAD_CYCLES_VALU", "SQ_ACTIVE_INST_VALU", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR",
              "TCP_TCC_READ_REQ_sum", "TCP_TOTAL_CACHE_ACCESSES_sum", "TCP_PENDING_STALL_CYCLES_sum", "TA_TA_BUSY_sum"):
        if k in c:
            o[k] = c[k]
    if "SQ_THREAD_CYCLES_VALU" in c and c.get("SQ_ACTIVE_INST_VALU"):
        o["lanes_per_valu_instruction"] = c["SQ_THREAD_CYCLES_VALU"] / c["SQ_ACTIVE_INST_VALU"]
    if "TCP_TCC_READ_REQ_sum" in c:
        o["l2_to_cu_bytes"] = c["TCP_TCC_READ_REQ_sum"] * 64.0
        if seconds:
            o["l2_to_cu_GBps"] = o["l2_to_cu_bytes"] / seconds / 1e9
        if c.get("TCP_TOTAL_CACHE_ACCESSES_sum"):
            o["l1_hit_rate"] = 1.0 - c["TCP_TCC_READ_REQ_sum"] / c["TCP_TOTAL_CACHE_ACCESSES_sum"]
            o["l1_access_bytes"] = c["TCP_TOTAL_CACHE_ACCESSES_sum"] * 64.0
    if "TA_TA_BUSY_sum" in c and seconds:
        o["ta_busy_share"] = c["TA_TA_BUSY_sum"] / 256.0 / (seconds * 2.4e9)   # per compute unit, against the kernel's cycles at 2.4 GHz
    if "SQ_INSTS_VALU" in c and seconds:
        o["valu_issue_frac"] = c["SQ_INSTS_VALU"] * 4.0 / 1024.0 / (seconds * 2.4e9)
    if "SQ_WAIT_ANY" in c and "SQ_WAVE_CYCLES" in c:
        o["wait_any_share_of_wave_cycles"] = c["SQ_WAIT_ANY"] / max(1.0, c["SQ_WAVE_CYCLES"])
    return o


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else "r02"
    src = os.path.join(ROOT, "gpurun_out", "prof_" + tag)
    dst = os.path.join(ROOT, "profiles")
    summary = {"tag": tag, "commit": subprocess.run(["git", "rev-parse", "--short", "HEAD"], cwd=ROOT, capture_output=True, text=True).stdout.strip(),
               "commands": {"headline": "RTGPU_RENDER_LANES=1 rocprofv3 [--kernel-trace --stats | --pmc <group>] -- python3 bench.py --no-cpu-baseline --no-extras --no-roofline "
                                        "(--steps 3 --warmup 1 for the trace, --steps 1 --warmup 0 for counters)",
                            "configs[1]": "rocprofv3 [...] -- python3 scripts/prof_c2.py (6 launches on primary rays, then 6 on bounce rays)"}}
    # ---- headline frame: trace_stage_kernel, per frame
    stats = newest(os.path.join(src, "trace", "*", "*_kernel_stats.csv"))[0]
    shutil.copy(stats, os.path.join(dst, tag + "_bench_kernel_stats.csv"))
    kt = kernel_time(stats, "trace_stage_kernel")
    frames_traced = 4   # --steps 3 --warmup 1
    c = {k: sum(v) for k, v in counters(sorted(glob.glob(os.path.join(src, "pmc_head_*"))), "trace_stage_kernel").items()}
    head = {"kernel": "trace_stage_kernel<kTraceSingle>", "kernel_trace": kt}
    if kt:
        head["launches_per_frame"] = kt["calls"] / frames_traced
        head["trace_stage_ms_per_frame"] = kt["total_ns"] / frames_traced / 1e6
        head.update(derived(c, kt["total_ns"] / frames_traced / 1e9))
        if "hbm_bytes" in head:
            head["hbm_bytes_per_frame"] = head.pop("hbm_bytes")
    summary["headline_trace_stage"] = head
    # ---- configs[1]: the closest-hit kernel, launches 0..5 primary rays, 6..11 bounce rays
    stats2 = newest(os.path.join(src, "trace_c2", "*", "*_kernel_stats.csv"))[0]
    shutil.copy(stats2, os.path.join(dst, tag + "_c2_kernel_stats.csv"))
    trace2 = newest(os.path.join(src, "trace_c2", "*", "*_kernel_trace.csv"))[0]
    durs = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in csv.DictReader(open(trace2)) if "trace_closest_steal_kernel" in r["Kernel_Name"]]
    lists = counters(sorted(glob.glob(os.path.join(src, "pmc_c2_*"))), "trace_closest_steal_kernel")
    for name, sl in (("c2_primary", slice(1, 6)), ("c2_bounce", slice(7, 12))):
        sec = sum(durs[sl]) / max(1, len(durs[sl])) / 1e9 if len(durs) >= 12 else None
        c = {k: sum(v[sl]) / max(1, len(v[sl])) for k, v in lists.items() if len(v) >= 12}
        e = {"kernel": "trace_closest_steal_kernel<false>", "kernel_trace_average_ns": sec * 1e9 if sec else None}
        e.update(derived(c, sec))
        if "hbm_bytes" in e:
            e["hbm_bytes_per_launch"] = e.pop("hbm_bytes")
        summary[name] = e
    # ---- configs[4]: trace_stage_kernel over two frames of the 49.8 M-triangle cluster
    st5 = newest(os.path.join(src, "trace_c5", "*", "*_kernel_stats.csv"))
    if st5:
        shutil.copy(st5[0], os.path.join(dst, tag + "_c5_kernel_stats.csv"))
        kt5 = kernel_time(st5[0], "trace_stage_kernel")
        c5 = {k: sum(v) / 2.0 for k, v in counters(sorted(glob.glob(os.path.join(src, "pmc_c5_*"))), "trace_stage_kernel").items()}
        e = {"kernel": "trace_stage_kernel<kTraceSingle>", "kernel_trace": kt5}
        if kt5:
            e["trace_stage_ms_per_frame"] = kt5["total_ns"] / 2.0 / 1e6
            e.update(derived(c5, kt5["total_ns"] / 2.0 / 1e9))
            if "hbm_bytes" in e:
                e["hbm_bytes_per_frame"] = e.pop("hbm_bytes")
        summary["c5_trace_stage"] = e
        summary["commands"]["configs[4]"] = "RTGPU_RENDER_LANES=1 rocprofv3 [...] -- python3 scripts/prof_c5.py (two frames; figures per frame)"
    # ---- configs[2] / configs[3]: trace_stage_kernel over frames of the interior scene (two 1080p frames / one 4K frame)
    for key, name, frames in (("interior_1080p_trace_stage", "i1080", 2.0), ("interior_4k_trace_stage", "i4k", 1.0)):
        sti = newest(os.path.join(src, "trace_" + name, "*", "*_kernel_stats.csv"))
        if not sti:
            continue
        shutil.copy(sti[0], os.path.join(dst, tag + "_" + name + "_kernel_stats.csv"))
        kti = kernel_time(sti[0], "trace_stage_kernel")
        ci = {k: sum(v) / frames for k, v in counters(sorted(glob.glob(os.path.join(src, "pmc_" + name + "_*"))), "trace_stage_kernel").items()}
        e = {"kernel": "trace_stage_kernel<kTraceSingle>", "kernel_trace": kti}
        if kti:
            e["trace_stage_ms_per_frame"] = kti["total_ns"] / frames / 1e6
            e.update(derived(ci, kti["total_ns"] / frames / 1e9))
            if "hbm_bytes" in e:
                e["hbm_bytes_per_frame"] = e.pop("hbm_bytes")
        summary[key] = e
    summary["commands"]["configs[2] / configs[3]"] = "RTGPU_RENDER_LANES=1 rocprofv3 [...] -- python3 scripts/prof_interior.py 1080p | 4k (two frames / one frame; figures per frame)"
    json.dump(summary, open(os.path.join(dst, tag + "_pmc_summary.json"), "w"), indent=1)
    print(json.dumps(summary, indent=1))


if __name__ == "__main__":
    main()
