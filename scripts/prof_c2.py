"""Profiling target for configs[1]: the closest-hit kernel on the blob's 1920x1080 primary rays (6 launches), then on the
same number of diffuse-bounce rays from the primary hit points (6 launches) -- the rays bench.py's "extras" use."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from raytracer_amd import abi, api, scenes

dev = torch.device("cuda", 0)
W, H = 1920, 1080
sc, cam_m, yfov = scenes.blob_scene(8)
cam = api.camera_from_transform(scenes.x_fov_degrees(yfov, W / H), cam_m)
scene = api.Scene(sc)
stream = torch.cuda.current_stream()
n = W * H
st = abi.make_settings(W, H, geometry_aa=1, material_aa=1, jitter_mode=abi.JITTER_CENTER)
d_o = torch.empty((n, 3), dtype=torch.float32, device=dev)
d_d = torch.empty((n, 3), dtype=torch.float32, device=dev)
api.generate_primary_device(cam, st, 0, d_o, d_d, stream)
hits = torch.empty((n, 5), dtype=torch.int32, device=dev)
for _ in range(6):
    scene.trace_rays_device(d_o, d_d, n, hits, stream)
torch.cuda.synchronize()
h = hits.cpu().numpy().view(np.uint32)
hit = h[:, 2] != 0xFFFFFFFF
o, d = d_o.cpu().numpy()[hit], d_d.cpu().numpy()[hit]
hp = (o + d * h[hit, 0].view(np.float32)[:, None]).astype(np.float32)
rng = np.random.RandomState(1)
so = np.tile(hp, (int(np.ceil(n / len(hp))), 1))[:n]
nn = so / np.linalg.norm(so, axis=1, keepdims=True)
so = (so + nn * 1e-3).astype(np.float32)
sd = rng.normal(size=so.shape)
sd /= np.linalg.norm(sd, axis=1, keepdims=True)
sd[(sd * nn).sum(1) < 0] *= -1
b_o, b_d = torch.from_numpy(so).to(dev), torch.from_numpy(sd.astype(np.float32)).to(dev)
for _ in range(6):
    scene.trace_rays_device(b_o, b_d, n, hits, stream)
torch.cuda.synchronize()
print("done")
