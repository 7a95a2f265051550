"""Developer probe: the page-locked host-array rate in a process that has rendered before (more streams alive)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import oracle
from raytracer_amd import abi, api, scenes

dev = torch.device("cuda", 0)
W, H = 1920, 1080
sc, cam_m, yfov = scenes.blob_scene(8)
cam = api.camera_from_transform(scenes.x_fov_degrees(yfov, W / H), cam_m)
o, d = oracle.generate_primary(cam, abi.make_settings(W, H, geometry_aa=1, material_aa=1, jitter_mode=abi.JITTER_CENTER))
po, pd, pout = api.host_empty(o.shape), api.host_empty(d.shape), api.host_empty(len(o), abi.HIT_DTYPE)
po[:] = o; pd[:] = d


def rate(g, label):
    g.trace_rays(po, pd, pout)
    ts = []
    for _ in range(10):
        t0 = time.perf_counter(); g.trace_rays(po, pd, pout); ts.append(time.perf_counter() - t0)
    print("%s: median %.2f ms, min %.2f, max %.2f" % (label, np.median(ts) * 1e3, min(ts) * 1e3, max(ts) * 1e3))


with api.Scene(sc) as g:
    rate(g, "fresh process")
    g.trace_rays(o, d)
    rate(g, "after a staged call (16 more streams)")
    if "--render" in sys.argv:
        sc2, cam2, yfov2 = scenes.blob_room_scene(6)
        with api.Scene(sc2) as g2:
            fb = torch.zeros((H, W, 3), dtype=torch.float32, device=dev)
            st = abi.make_settings(W, H, geometry_aa=2, material_aa=2, max_path_length=9, seed=1)
            api.Renderer().render_device(g2, api.camera_from_transform(scenes.x_fov_degrees(yfov2, W / H), cam2), st, fb)
            torch.cuda.synchronize()
            rate(g, "after a render on another scene (alive)")
        rate(g, "after that scene is gone")
    x = torch.zeros(1 << 28, device=dev)   # 1 GiB
    torch.cuda.synchronize()
    rate(g, "with a torch tensor of 1 GiB alive")
