"""bench.py end to end: the one-line JSON contract at N = 1 (reduced frame size), and the multi-rank flow (barriers,
max over ranks, squares per rank + gather) rehearsed with two ranks sharing this box's GPU over gloo -- the measured
multi-GPU configuration is one rank per GPU over RCCL, which needs a multi-GPU node."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KEYS = {"metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
        "dtype", "data", "config"}
SMALL = ["--width", "960", "--height", "540", "--gaa", "4", "--maa", "4", "--subdiv", "6", "--no-c3"]


def check_line(out, n_gpus, steps):
    lines = [l for l in out.strip().splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert KEYS <= set(d) and d["n_gpus"] == n_gpus and d["steps"] == steps and d["unit"] == "Mrays/s"
    assert d["value"] > 100 and d["higher_is_better"] is True and d["scaling"] == "strong" and d["vs_baseline"] is None
    assert "workload" in d["config"] and "model" not in d["config"] and "%d GPU(s)" % n_gpus in d["config"]["workload"]
    assert d["config"]["rays_per_frame"] > 1e7 and d["config"]["non_finite_pixels"] < 10
    return d


def check_roofline(r):
    """bound = the resource that binds (VALU issue, from committed counters of the same kernel and workload) with frac < 1, or --
    without such counters -- the algorithmic bytes against the HBM peak, which a cache-resident scene may exceed; the
    algorithmic rate against the L2 peak stays below 1 either way."""
    assert r["bound"] in ("valu_issue", "hbm") and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9
    assert {"achieved", "peak", "unit", "traffic", "algorithmic_GBps", "frac_of_l2_peak", "frac_of_hbm_peak"} <= set(r)
    assert 0 < r["frac_of_l2_peak"] < 1
    if r["bound"] == "valu_issue":
        assert 0 < r["frac"] < 1 and r["peak"] == 614.4 and "committed profile" in r["counters"]["source"] and r["traffic"] > 0
    else:
        assert r["peak"] == 8000.0 and r["frac"] > 0 and "counters" not in r


def test_bench_single_gpu_line():
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1", "--no-c5"] + SMALL,
                       capture_output=True, text=True, timeout=400, cwd=ROOT)
    assert p.returncode == 0, p.stderr[-2000:]
    d = check_line(p.stdout, 1, 2)
    check_roofline(d["roofline"])
    assert d["roofline"]["kernel"].startswith("trace_stage_kernel") and d["roofline"]["extension"]["rays"] > 0
    assert d["cpu_baseline"]["value"] > 0 and d["cpu_baseline"]["kind"] == "port" and d["cpu_baseline"]["gpu_tile_vs_oracle_rel_rmse"] < 1e-3
    ex = d["extras"]
    check_roofline(ex["configs[1]"]["primary"]["roofline"])
    assert "counters" not in d["roofline"]   # a reduced frame does not quote the committed counters of the full one
    assert ex["configs[0]"]["identical_to_oracle"] is True and ex["configs[0]"]["cpu_baseline"]["value"] > 0
    check_roofline(ex["configs[0]"]["roofline"])
    check_roofline(ex["configs[1]"]["bounce"]["roofline"])
    check_roofline(ex["configs[2]"]["roofline"])
    assert ex["configs[2]"]["ms_per_frame"] > 0 and ex["configs[2]"]["rays_per_frame"] > 1e8


def test_bench_two_ranks_rehearsal_over_gloo():
    env = dict(os.environ, RTGPU_BENCH_BACKEND="gloo")
    p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", "29517", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1"] + SMALL,
                       capture_output=True, text=True, timeout=400, cwd=ROOT, env=env)
    assert p.returncode == 0, (p.stdout + p.stderr)[-3000:]
    d = check_line(p.stdout, 2, 2)
    assert "cpu_baseline" not in d and "extras" not in d and "rehearsal" in d["config"]["gather"]
    check_roofline(d["roofline"])   # rank 0 measures the kernels' roofline on the whole frame, whatever the number of ranks
    assert "rank 0" in d["roofline"]["measured"]
