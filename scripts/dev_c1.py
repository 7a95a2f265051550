"""Developer probe: config 1 (101 mesh instances + 16 analytic spheres, 640x480, primary + shadow rays)
device-side timing of the trace kernels, with and without the spheres and as a soup."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from raytracer_amd import scenes, abi, api
dev = torch.device("cuda", 0); stream = torch.cuda.current_stream()
W, H = 640, 480
for label, kw in (("instanced + spheres (reference-order top level)", dict()), ("instanced meshes only", dict(with_spheres=False)), ("soup", dict(with_spheres=False, soup=True))):
    sc, cam_m, yfov = scenes.c1_scene(seed=0, **kw)
    g = api.Scene(sc)
    cam = api.camera_from_transform(scenes.x_fov_degrees(yfov, W / H), cam_m)
    st = abi.make_settings(W, H, geometry_aa=1, material_aa=1, jitter_mode=abi.JITTER_CENTER)
    n = W * H
    d_o = torch.empty((n, 3), dtype=torch.float32, device=dev); d_d = torch.empty((n, 3), dtype=torch.float32, device=dev)
    api.generate_primary_device(cam, st, 0, d_o, d_d, stream)
    hits = torch.empty((n, 5), dtype=torch.int32, device=dev)
    for _ in range(5): g.trace_rays_device(d_o, d_d, n, hits, stream)
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record(stream)
    for _ in range(50): g.trace_rays_device(d_o, d_d, n, hits, stream)
    e1.record(stream); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 50
    print(f"{label}: tris={sc.n_triangles()} primary {ms*1e3:.1f} us {n/ms/1e3:.0f} Mrays/s")
    g.close()
