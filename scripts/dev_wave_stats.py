"""Developer probe: wave-schedule statistics of the work-stealing traversal (needs the
`make -C raytracer_amd/csrc dev` library: RTGPU_LIB=raytracer_amd/librtgpu_dev.so)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from raytracer_amd import scenes, abi, api

dev = torch.device("cuda", 0)
stream = torch.cuda.current_stream()
W, H = 1920, 1080

def stats(g, o, d, label, show_timeline=False):
    n = o.shape[0] // 64 * 64
    o, d = o[:n], d[:n]
    d_o = torch.from_numpy(np.ascontiguousarray(o)).to(dev); d_d = torch.from_numpy(np.ascontiguousarray(d)).to(dev)
    cnt = torch.zeros((n, 2), dtype=torch.int32, device=dev)
    for _ in range(3):
        g.trace_counts_device(d_o, d_d, n, cnt, stream); torch.cuda.synchronize()
    c = cnt.cpu().numpy().view(np.uint32).astype(np.int64)
    # the launch maps position p -> ray interleave8(p); undo it to find each wave's lanes
    nb8 = (n >> 3) & ~7
    p = np.arange(n)
    ray = np.where(p < nb8 * 8, (((p >> 6) + ((p >> 3) & 7) * (nb8 >> 3)) << 3) + (p & 7), p)
    w = c[ray].reshape(n // 64, 64, 2)
    nw = n // 64
    boxes = w[:, :, 0].sum()
    tris = w[:, 12:, 1].sum() * 64 / 52
    hi56, hi48, lo32, lo16 = w[:, 8:12, 1].sum(0).astype(np.float64)
    outer, steal, nodeit, leafit, busy, stolen = w[:, :6, 1].sum(0).astype(np.float64)
    print(f"{label}: rays={n} waves={nw} boxes/ray={boxes/n:.2f} tris/ray~{tris/n:.2f}")
    print(f"   per wave: iterations={outer/nw:.1f} steal rounds={steal/nw:.1f} stolen entries={stolen/nw:.1f} node steps={nodeit/nw:.1f} leaf steps={leafit/nw:.1f} busy lanes per iteration={busy/max(outer,1):.1f}")
    print(f"   share of iterations with >= 56 busy lanes {hi56/outer:.2f}, >= 48: {hi48/outer:.2f}, < 32: {lo32/outer:.2f}, < 16: {lo16/outer:.2f}")
    print(f"   boxes per node step={boxes/nodeit:.1f} (of 256) triangles per leaf step~{tris/leafit:.1f} (of 64)")
    t0 = w[:, 6, 1]; t1 = w[:, 7, 1]
    base = t0.min()
    t0 = (t0 - base) / 100.0; t1 = (t1 - base) / 100.0   # us (100 MHz clock)
    dur = t1 - t0
    print(f"   span {t1.max():.1f} us; wave life mean {dur.mean():.1f} us p50 {np.median(dur):.1f} p99 {np.percentile(dur,99):.1f} max {dur.max():.1f}; last start {t0.max():.1f} us")
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
ro, rd = scenes.random_rays(2_000_000, (-1.6, -1.6, -1.6), (1.6, 1.6, 1.6), seed=2)
stats(g, ro, rd, "random incoherent")
# diffuse-bounce rays from the primary hit points (as bench.py's configs[1] makes them)
hits = torch.empty((n, 5), dtype=torch.int32, device=dev)
g.trace_rays_device(d_o, d_d, n, hits, stream); torch.cuda.synchronize()
h = hits.cpu().numpy().view(np.uint32)
hit = h[:, 2] != 0xFFFFFFFF
t = h[hit, 0].view(np.float32)
hp = (o[hit] + d[hit] * t[:, None]).astype(np.float32)
rng = np.random.RandomState(1)
so = np.tile(hp, (int(np.ceil(n / len(hp))), 1))[:n]
nn = so / np.linalg.norm(so, axis=1, keepdims=True)
so = (so + nn * 1e-3).astype(np.float32)
sd = rng.normal(size=so.shape); sd /= np.linalg.norm(sd, axis=1, keepdims=True)
sd[(sd * nn).sum(1) < 0] *= -1
stats(g, so, sd.astype(np.float32), "bounce rays")
