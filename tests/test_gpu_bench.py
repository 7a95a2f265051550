"""bench.py end to end: the one-line JSON contract at N = 1, and the multi-rank flow (barriers, max over
ranks, sharded frame + gather) rehearsed with two ranks sharing this box's GPU over gloo -- the measured
multi-GPU configuration is one rank per GPU over RCCL, which needs a multi-GPU node."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KEYS = {"metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
        "dtype", "data", "config", "roofline"}


def check_line(out, n_gpus, steps):
    lines = [l for l in out.strip().splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert KEYS <= set(d) and d["n_gpus"] == n_gpus and d["steps"] == steps and d["unit"] == "Mrays/s"
    assert d["value"] > 100 and d["higher_is_better"] is True and d["scaling"] == "weak" and d["vs_baseline"] is None
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["peak"] == 8000.0 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9
    assert "workload" in d["config"] and "model" not in d["config"]
    assert d["frame"]["ms_per_frame"] > 0 and d["frame"]["rays_per_frame"] > 1e8
    return d


def test_bench_single_gpu_line():
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "5", "--warmup", "1", "--no-cpu-baseline"],
                       capture_output=True, text=True, timeout=300, cwd=ROOT)
    assert p.returncode == 0, p.stderr[-2000:]
    check_line(p.stdout, 1, 5)


def test_bench_two_ranks_rehearsal_over_gloo():
    env = dict(os.environ, RTGPU_BENCH_BACKEND="gloo")
    p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", "29517", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "1"],
                       capture_output=True, text=True, timeout=400, cwd=ROOT, env=env)
    assert p.returncode == 0, (p.stdout + p.stderr)[-3000:]
    d = check_line(p.stdout, 2, 4)
    assert "cpu_baseline" not in d and "2 GPU(s)" in d["frame"]["workload"]
