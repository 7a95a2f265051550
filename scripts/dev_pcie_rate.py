"""Developer measurement: the host-pointer entry point (rays and hits cross PCIe) on the bench workload."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch  # noqa: F401
import oracle
from raytracer_amd import abi, api, scenes

sc, cam_m, yfov = scenes.blob_scene(8)
W, H = 1920, 1080
cam = api.camera_from_transform(scenes.x_fov_degrees(yfov, W / H), cam_m)
o, d = oracle.generate_primary(cam, abi.make_settings(W, H, geometry_aa=1, material_aa=1, jitter_mode=abi.JITTER_CENTER))
with api.Scene(sc) as g:
    g.trace_rays(o, d)
    ts = []
    for _ in range(10):
        t0 = time.perf_counter(); g.trace_rays(o, d); ts.append(time.perf_counter() - t0)
    t = float(np.median(ts))
    print("rtgpu_trace_closest (host arrays, %d rays): %.2f ms = %.0f Mrays/s" % (len(o), t * 1e3, len(o) / t / 1e6))
