"""Developer probe: wave-schedule statistics of the persistent-wave traversal (needs the
`make -C raytracer_amd/csrc dev` library: RTGPU_LIB=raytracer_amd/librtgpu_dev.so)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from raytracer_amd import scenes, abi, api

dev = torch.device("cuda", 0)
stream = torch.cuda.current_stream()
W, H = 1920, 1080
EXTRA = 40000  # rows behind the ray counts that receive 8 words per wave

def run(g, o, d):
    n = o.shape[0]
    d_o = torch.from_numpy(np.ascontiguousarray(o)).to(dev); d_d = torch.from_numpy(np.ascontiguousarray(d)).to(dev)
    cnt = torch.zeros((n + EXTRA, 2), dtype=torch.int32, device=dev)
    for _ in range(3):
        cnt.zero_()
        g.trace_counts_device(d_o, d_d, n, cnt, stream); torch.cuda.synchronize()
    c = cnt.cpu().numpy().view(np.uint32)
    w = c[n:].reshape(-1, 8).astype(np.int64)
    w = w[w[:, 7] != 0]
    return n, c[:n].astype(np.int64), w

def stats(g, o, d, label, show_timeline=False):
    n, c, w = run(g, o, d)
    nw = len(w)
    boxes = c[:, 0].sum()
    outer, steal, nodeit, leafit, busy, stolen = w[:, :6].sum(0).astype(np.float64)
    print(f"{label}: rays={n} waves={nw} lane node visits/ray={boxes/2/n:.2f} tris/ray={c[:,1].sum()/n:.2f}")
    print(f"   per wave: iterations={outer/nw:.1f} steal rounds={steal/nw:.1f} stolen entries={stolen/nw:.1f} node steps={nodeit/nw:.1f} leaf steps={leafit/nw:.1f} busy lanes per iteration={busy/max(outer,1):.1f}")
    print(f"   node-step SIMT efficiency={boxes/2/(nodeit*64):.3f} leaf-step SIMT efficiency={c[:,1].sum()/(leafit*64):.3f}")
    t0 = w[:, 6]; t1 = w[:, 7]
    base = t0.min()
    t0 = (t0 - base) / 100.0; t1 = (t1 - base) / 100.0   # us (100 MHz clock)
    dur = t1 - t0
    print(f"   span {t1.max():.1f} us; wave life mean {dur.mean():.1f} us p50 {np.median(dur):.1f} p99 {np.percentile(dur,99):.1f} max {dur.max():.1f}; last start {t0.max():.1f} us; first end {t1.min():.1f} us")
    if show_timeline:
        for a in np.arange(0, t1.max() + 20, 20.0):
            act = ((t0 < a + 20) & (t1 > a)).sum()
            print(f"   t={a:6.0f} us active waves {act:6d}")

sc, cam_m, yfov = scenes.blob_scene(8)
g = api.Scene(sc)
cam = api.camera_from_transform(scenes.x_fov_degrees(yfov, W / H), cam_m)
st = abi.make_settings(W, H, geometry_aa=1, material_aa=1, jitter_mode=abi.JITTER_CENTER)
n = W * H
d_o = torch.empty((n, 3), dtype=torch.float32, device=dev); d_d = torch.empty((n, 3), dtype=torch.float32, device=dev)
api.generate_primary_device(cam, st, 0, d_o, d_d, stream); torch.cuda.synchronize()
o = d_o.cpu().numpy(); d = d_d.cpu().numpy()
stats(g, o, d, "primary rows", True)
idx = np.arange(n).reshape(H // 8, 8, W // 8, 8).transpose(0, 2, 1, 3).reshape(-1)
stats(g, o[idx], d[idx], "primary 8x8 tiles")
ro, rd = scenes.random_rays(2_000_000, (-1.6, -1.6, -1.6), (1.6, 1.6, 1.6), seed=2)
stats(g, ro, rd, "random incoherent")
