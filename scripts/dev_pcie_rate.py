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
    out = np.empty(len(o), dtype=abi.HIT_DTYPE)
    out[:] = 0
    g.trace_rays(o, d, out)
    ts = []
    for _ in range(10):
        t0 = time.perf_counter(); g.trace_rays(o, d, out); ts.append(time.perf_counter() - t0)
    t = float(np.median(ts))
    print("the same into a result array the caller keeps: %.2f ms = %.0f Mrays/s" % (t * 1e3, len(o) / t / 1e6))
    o4, d4 = np.tile(o, (4, 1)), np.tile(d, (4, 1))
    out4 = np.zeros(len(o4), dtype=abi.HIT_DTYPE)
    g.trace_rays(o4, d4, out4)
    ts = []
    for _ in range(5):
        t0 = time.perf_counter(); g.trace_rays(o4, d4, out4); ts.append(time.perf_counter() - t0)
    t = float(np.median(ts))
    print("four times the rays (%d): %.2f ms = %.0f Mrays/s" % (len(o4), t * 1e3, len(o4) / t / 1e6))
    po, pd = api.host_empty(o.shape), api.host_empty(d.shape)
    pout = api.host_empty(len(o), abi.HIT_DTYPE)
    po[:] = o; pd[:] = d; pout[:] = 0
    g.trace_rays(po, pd, pout)
    assert np.array_equal(pout.view(np.uint8), out.view(np.uint8)), "page-locked arrays give different hits"
    ts = []
    for _ in range(10):
        t0 = time.perf_counter(); g.trace_rays(po, pd, pout); ts.append(time.perf_counter() - t0)
    t = float(np.median(ts))
    print("page-locked arrays (rtgpu_host_alloc): %.2f ms = %.0f Mrays/s" % (t * 1e3, len(o) / t / 1e6))
