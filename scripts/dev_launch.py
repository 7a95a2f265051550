"""Developer probe: launch-rate floor and scaling of the closest-hit kernel."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from raytracer_amd import scenes, abi, api
dev = torch.device("cuda", 0); stream = torch.cuda.current_stream()
W, H = 1920, 1080
sc, cam_m, yfov = scenes.blob_scene(8)
g = api.Scene(sc)
cam = api.camera_from_transform(scenes.x_fov_degrees(yfov, W / H), cam_m)
st = abi.make_settings(W, H, geometry_aa=1, material_aa=1, jitter_mode=abi.JITTER_CENTER)
n = W * H
d_o = torch.empty((n, 3), dtype=torch.float32, device=dev); d_d = torch.empty((n, 3), dtype=torch.float32, device=dev)
api.generate_primary_device(cam, st, 0, d_o, d_d, stream); torch.cuda.synchronize()
def timeit(o, d, reps=10):
    m = o.shape[0]
    hits = torch.empty((m, 5), dtype=torch.int32, device=dev)
    for _ in range(3): g.trace_rays_device(o, d, m, hits, stream)
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record(stream)
    for _ in range(reps): g.trace_rays_device(o, d, m, hits, stream)
    e1.record(stream); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
print("full frame          : %.3f ms" % timeit(d_o, d_d))
print("all-miss (d = -d)   : %.3f ms" % timeit(d_o, (-d_d).contiguous()))
print("4x frame            : %.3f ms" % timeit(d_o.repeat(4, 1), d_d.repeat(4, 1)))
print("rows 270..810 (half): %.3f ms" % timeit(d_o[270 * W:810 * W].contiguous(), d_d[270 * W:810 * W].contiguous()))
perm = torch.randperm(n, device=dev)
print("shuffled rays       : %.3f ms" % timeit(d_o[perm].contiguous(), d_d[perm].contiguous()))
# 8x8 tiled order
idx = torch.arange(n, device=dev).reshape(H // 8, 8, W // 8, 8).permute(0, 2, 1, 3).reshape(-1)
print("8x8 tiled order     : %.3f ms" % timeit(d_o[idx].contiguous(), d_d[idx].contiguous()))
# only hit rays
hits = torch.empty((n, 5), dtype=torch.int32, device=dev); g.trace_rays_device(d_o, d_d, n, hits, stream); torch.cuda.synchronize()
m = hits[:, 2] != -1
print("hit rays only (%d, tiled): %.3f ms" % (int(m.sum()), timeit(d_o[idx][m[idx]].contiguous(), d_d[idx][m[idx]].contiguous())))
# separate intra-wave coherence from inter-wave locality
def order_time(name, idx):
    print("%-34s: %.3f ms" % (name, timeit(d_o[idx].contiguous(), d_d[idx].contiguous())))
rows = torch.arange(n, device=dev).reshape(-1, 64)
order_time("64x1 spans, spans shuffled", rows[torch.randperm(rows.shape[0], device=dev)].reshape(-1))
tiles = torch.arange(n, device=dev).reshape(H // 8, 8, W // 8, 8).permute(0, 2, 1, 3).reshape(-1, 64)
order_time("8x8 tiles, tiles shuffled", tiles[torch.randperm(tiles.shape[0], device=dev)].reshape(-1))
t4 = torch.arange(n, device=dev).reshape(H // 4, 4, W // 16, 16).permute(0, 2, 1, 3).reshape(-1, 64)
order_time("16x4 tiles in order", t4.reshape(-1))
order_time("16x4 tiles shuffled", t4[torch.randperm(t4.shape[0], device=dev)].reshape(-1))
order_time("row-major reversed", torch.arange(n - 1, -1, -1, device=dev))
# interleave: wave k takes pixels k, k+nw, k+2nw... (maximally incoherent lanes but same set)
order_time("lane-strided (transpose)", torch.arange(n, device=dev).reshape(64, -1).t().reshape(-1))
cnt = torch.zeros((n, 2), dtype=torch.int32, device=dev)
g.trace_counts_device(d_o, d_d, n, cnt, stream); torch.cuda.synchronize()
steps = (cnt[:, 0] // 2 + cnt[:, 1])
for thr in (50, 100, 200, 1000):
    keep = steps < thr
    k = int(keep.sum())
    print("rays with < %4d steps (%7d rays, row order): %.3f ms" % (thr, k, timeit(d_o[keep].contiguous(), d_d[keep].contiguous())))
heavy = steps >= 100
print("only rays with >= 100 steps (%d rays): %.3f ms" % (int(heavy.sum()), timeit(d_o[heavy].contiguous(), d_d[heavy].contiguous())))
srt = torch.argsort(steps, descending=True)
print("sorted by steps descending: %.3f ms" % timeit(d_o[srt].contiguous(), d_d[srt].contiguous()))
