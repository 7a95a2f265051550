"""CPU tests of the host-side logic: scene generators, band partition, and the multi-rank gather
path on the gloo backend (world size 2)."""
import os
import socket
import subprocess
import sys

import numpy as np

from raytracer_amd import abi, distributed, scenes

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_scene_generators():
    m = scenes.uv_sphere(32, 16)
    assert m.indices.shape == (960, 3) and m.vertices.shape[0] == 482
    sc, cam_m, yfov = scenes.c1_scene()
    assert len(sc.instances) == 117 and sc.n_triangles() == 960
    soup, _, _ = scenes.c1_scene(soup=True, with_spheres=False)
    assert len(soup.instances) == 1 and soup.n_triangles() == 101 * 960
    v, f = scenes.icosphere(3)
    assert f.shape[0] == 20 * 4 ** 3 and v.shape[0] == 10 * 4 ** 3 + 2
    box, _, _ = scenes.box_scene()
    assert box.meshes[0].tri_material is not None and box.meshes[0].tri_material.max() < len(box.materials)
    small, _, _ = scenes.interior_scene(detail=0.02)
    assert small.n_triangles() > 1000


def test_band_partition_covers_every_row_once():
    for H in (1, 7, 48, 1080, 2160):
        for world in (1, 2, 4, 8):
            seen = np.zeros(H, dtype=int)
            for r in range(world):
                for y0, y1 in distributed.rank_bands(H, world, r):
                    assert 0 <= y0 < y1 <= H
                    seen[y0:y1] += 1
            assert np.all(seen == 1)
            rows = [sum(y1 - y0 for y0, y1 in distributed.rank_bands(H, world, r)) for r in range(world)]
            assert max(rows) == distributed.padded_rows(H, world)
            if H >= world * 8:
                assert max(rows) - min(rows) <= distributed.band_height(H, world)


WORKER = r"""
import os, sys
sys.path.insert(0, {root!r})
import numpy as np, torch, torch.distributed as dist
from raytracer_amd import distributed
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo")
W, H = 37, 53
def value(y0, y1):
    ys = np.arange(y0, y1, dtype=np.float32)[:, None, None]
    xs = np.arange(W, dtype=np.float32)[None, :, None]
    cs = np.arange(3, dtype=np.float32)[None, None, :]
    return (ys * 1000 + xs * 3 + cs).astype(np.float32) * np.ones((y1 - y0, W, 3), np.float32)
def render_band(y0, y1, out):
    out.copy_(torch.from_numpy(value(y0, y1)))
frame = distributed.render_sharded(render_band, W, H, rank, world, torch.device("cpu"))
if rank == 0:
    want = distributed.assemble_reference(H, W, world, value)
    assert np.array_equal(frame.numpy(), want) and np.array_equal(want, value(0, H))
    print("GATHER_OK")
else:
    assert frame is None
dist.barrier()
dist.destroy_process_group()
"""


def test_sharded_render_gather_gloo_world2(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER.format(root=ROOT))
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    outs = [p.communicate(timeout=180)[0].decode() for p in procs]
    assert all(p.returncode == 0 for p in procs), outs
    assert "GATHER_OK" in outs[0]
