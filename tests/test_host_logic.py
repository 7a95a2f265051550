"""CPU tests of the host-side logic: scene generators, the square order of the multi-GPU path, and the multi-rank
gather on the gloo backend (world size 2)."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

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


def _pattern(W, H):
    ys = np.arange(H, dtype=np.float32)[:, None, None]
    xs = np.arange(W, dtype=np.float32)[None, :, None]
    cs = np.arange(3, dtype=np.float32)[None, None, :]
    return (ys * 4096 + xs * 3 + cs + 1).astype(np.float32) * np.ones((H, W, 3), np.float32)


def test_square_order_is_a_permutation_for_every_world_size():
    """Cutting a frame into squares shared out over G ranks and writing the ranks' compact buffers back gives the frame
    bit for bit, for G = 1, 2, 4, 8, whole frames and sub-tiles, squares that stick out of the tile included; every
    pixel of the tile is owned exactly once; sizes are the ones the library reports."""
    for (W, H, tile, T) in ((37, 53, None, 8), (64, 64, None, 32), (1920, 1080, None, 32), (100, 70, (10, 5, 93, 64), 16), (5, 3, None, 32)):
        frame = _pattern(W, H)
        st = abi.make_settings(W, H, tile=tile)
        xs, ys, xe, ye = (0, 0, W, H) if tile is None else tile
        for world in (1, 2, 4, 8):
            bufs = [distributed.compact_from_frame(frame, st, T, r, world) for r in range(world)]
            for r, b in enumerate(bufs):
                assert b.size == distributed.tile_buffer_floats(st, T, r, world)
            assert bufs[0].size == max(b.size for b in bufs)                 # rank 0's buffer is the largest: the gather's stride
            assert sum(int((b.reshape(-1, 3)[:, 2] != 0).sum()) for b in bufs) == (xe - xs) * (ye - ys)   # each pixel once (values are > 0)
            background = np.full((H, W, 3), -1.0, np.float32)
            out = distributed.scatter_to_frame(bufs, st, T, background.copy())
            assert np.array_equal(out[ys:ye, xs:xe], frame[ys:ye, xs:xe])
            mask = np.ones((H, W), bool)
            mask[ys:ye, xs:xe] = False
            assert (out[mask] == -1.0).all()                                  # nothing outside the tile is written


WORKER = r"""
import os, sys
sys.path.insert(0, {root!r})
import numpy as np, torch, torch.distributed as dist
from raytracer_amd import abi, distributed
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo")
W, H, T = 37, 53, 8
ys = np.arange(H, dtype=np.float32)[:, None, None]
xs = np.arange(W, dtype=np.float32)[None, :, None]
cs = np.arange(3, dtype=np.float32)[None, None, :]
frame = (ys * 1000 + xs * 3 + cs).astype(np.float32) * np.ones((H, W, 3), np.float32)
st = abi.make_settings(W, H)
local = torch.from_numpy(distributed.compact_from_frame(frame, st, T, rank, world))   # what this rank "rendered"
got = distributed.gather_compact(local, st, T, rank, world)
if rank == 0:
    assert np.array_equal(got, frame)
    print("GATHER_OK")
else:
    assert got is None
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


@pytest.mark.parametrize("kind", [0, 1, 2])
def test_quantised_nodes_contain_the_exact_child_boxes(tmp_path, kind):
    """The 64-byte nodes the kernels fetch (8-bit planes on the node's own grid): tests/native/check_quantised_nodes.cpp
    builds a tree over 200 000 boxes (small boxes / flat boxes at large coordinates / some non-finite boxes), collapses and
    quantises it with the library's own code and checks every decoded plane against the exact one."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "check_q")
    subprocess.run(["g++", "-O2", "-std=c++17", "-pthread", "-I" + os.path.join(root, "raytracer_amd", "csrc"), "-I" + os.path.join(root, "include"),
                    "-o", exe, os.path.join(root, "tests", "native", "check_quantised_nodes.cpp")], check=True)
    r = subprocess.run([exe, "200000", str(kind + 1), str(kind)], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
    assert r.stdout.startswith("ok:")


@pytest.mark.parametrize("distance", [1, 200, 10000, 1000000])
def test_quantised_decode_never_rejects_a_box_a_ray_passes_through(tmp_path, distance):
    """The kernels' box test on quantised nodes, restated on the host in fp32 (tests/native/check_quantised_decode.cpp): rays
    aimed at corners, edges and faces of the child boxes of a tree over 20 000 boxes, from `distance` units away.  The widened
    decode (taken when a wave holds a ray farther than 256 units from its tree, and by the instanced and per-lane walks)
    accepts every ray that passes through the exact box at any distance; the plain decode accepts every ray through the
    box without the reference's 1e-4 padding as long as the rays start within its range (checked for distance <= 240)."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "check_qd")
    subprocess.run(["g++", "-O2", "-std=c++17", "-ffp-contract=off", "-pthread", "-I" + os.path.join(root, "raytracer_amd", "csrc"),
                    "-I" + os.path.join(root, "include"), "-o", exe, os.path.join(root, "tests", "native", "check_quantised_decode.cpp")], check=True)
    r = subprocess.run([exe, "20000", "3", str(distance)], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
    assert r.stdout.startswith("ok:")


def test_launch_uniform_division_is_exact(tmp_path):
    """rt_fastdiv.h (one multiply and a shift with a host-prepared factor, used by the integrator's index arithmetic): equal to
    the integer division for every numerator below 2^31 that was tried -- multiples of the divisor and their neighbours, the
    ends of the range, random values -- for every divisor up to 70 000, powers of two and their neighbours, and 200 000 random
    divisors."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "check_fastdiv")
    subprocess.run(["g++", "-O2", "-std=c++17", "-I" + os.path.join(root, "raytracer_amd", "csrc"), "-o", exe,
                    os.path.join(root, "tests", "native", "check_fastdiv.cpp")], check=True)
    r = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
    assert r.stdout.startswith("ok:")
