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

def stats(g, o, d, label):
    n = o.shape[0]
    d_o = torch.from_numpy(np.ascontiguousarray(o)).to(dev); d_d = torch.from_numpy(np.ascontiguousarray(d)).to(dev)
    cnt = torch.zeros((n, 2), dtype=torch.int32, device=dev)
    g.trace_counts_device(d_o, d_d, n, cnt, stream); torch.cuda.synchronize()
    c = cnt.cpu().numpy().astype(np.int64)
    nw = n // 64
    w = c[: nw * 64].reshape(nw, 64, 2)
    boxes = w[:, :, 0].sum()
    s = w[:, :6, 1].sum(0).astype(np.float64)
    outer, steal, nodeit, leafit, busy, stolen = s
    print(f"{label}: waves={nw} lane node visits/ray={boxes/2/n:.2f}")
    print(f"   per wave: outer={outer/nw:.1f} steal rounds={steal/nw:.1f} stolen entries={stolen/nw:.1f} node iters={nodeit/nw:.1f} leaf iters={leafit/nw:.1f} busy lanes at outer={busy/max(outer,1):.1f}")
    print(f"   node-loop SIMT efficiency={boxes/2/(nodeit*64):.3f}  node iters per outer={nodeit/outer:.1f} leaf iters per outer={leafit/outer:.2f}")

sc, cam_m, yfov = scenes.blob_scene(8)
g = api.Scene(sc)
cam = api.camera_from_transform(scenes.x_fov_degrees(yfov, W / H), cam_m)
st = abi.make_settings(W, H, geometry_aa=1, material_aa=1, jitter_mode=abi.JITTER_CENTER)
n = W * H
d_o = torch.empty((n, 3), dtype=torch.float32, device=dev); d_d = torch.empty((n, 3), dtype=torch.float32, device=dev)
api.generate_primary_device(cam, st, 0, d_o, d_d, stream); torch.cuda.synchronize()
o = d_o.cpu().numpy(); d = d_d.cpu().numpy()
stats(g, o, d, "primary rows")
# 8x8 tiles
idx = np.arange(n).reshape(H // 8, 8, W // 8, 8).transpose(0, 2, 1, 3).reshape(-1)
stats(g, o[idx], d[idx], "primary 8x8 tiles")
hits = g.trace_rays(o, d)
m = hits["inst"] != abi.INVALID_ID
stats(g, o[m][: (m.sum() // 64) * 64], d[m][: (m.sum() // 64) * 64], "primary hits only")
ro, rd = scenes.random_rays(2_000_000 // 64 * 64, (-1.6, -1.6, -1.6), (1.6, 1.6, 1.6), seed=2)
stats(g, ro, rd, "random incoherent")
