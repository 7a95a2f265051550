"""Developer probe: how long one rank's share of the headline frame takes for G = 1, 2, 4, 8 ranks, measured on one GPU
(render_tiles_device with rank r of G) -- the strong-scaling ceiling of the render itself, before any exchange."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from raytracer_amd import abi, api, scenes

W, H, G_AA, M_AA = 3840, 2160, 8, 8
if len(sys.argv) > 1:
    W, H = int(sys.argv[1]), int(sys.argv[2])
dev = torch.device("cuda", 0)
sc, cam_m, yfov = scenes.blob_room_scene(8)
cam = api.camera_from_transform(scenes.x_fov_degrees(yfov, W / H), cam_m)
scene = api.Scene(sc)
st = abi.make_settings(W, H, geometry_aa=G_AA, material_aa=M_AA, max_path_length=9, seed=1)
r = api.Renderer()
base = None
for world in (1, 2, 4, 8):
    worst, total = 0.0, 0.0
    for rank in sorted({0, world // 2, world - 1}):
        buf = torch.empty(max(api.tile_buffer_floats(st, 32, rank, world), 1), dtype=torch.float32, device=dev)
        r.render_tiles_device(scene, cam, st, 32, rank, world, buf)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3):
            r.render_tiles_device(scene, cam, st, 32, rank, world, buf)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / 3 * 1e3
        worst = max(worst, ms)
        print("  G=%d rank %d: %.2f ms" % (world, rank, ms), flush=True)
    if base is None:
        base = worst
    print("G=%d: slowest sampled rank %.2f ms, ideal %.2f ms, efficiency of the render alone %.3f" % (world, worst, base / world, base / world / worst), flush=True)
