"""Developer probe: boxes and triangles tested per ray (every ray traversed on its own) on configs[1] primary rays, for RTGPU_LIB."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from raytracer_amd import abi, api, scenes

dev = torch.device("cuda", 0)
stream = torch.cuda.current_stream()
W, H = 1920, 1080
sc, cam_m, yfov = scenes.blob_scene(8)
cam = api.camera_from_transform(scenes.x_fov_degrees(yfov, W / H), cam_m)
scene = api.Scene(sc)
n = W * H
st = abi.make_settings(W, H, geometry_aa=1, material_aa=1, jitter_mode=abi.JITTER_CENTER)
d_o = torch.empty((n, 3), dtype=torch.float32, device=dev)
d_d = torch.empty((n, 3), dtype=torch.float32, device=dev)
api.generate_primary_device(cam, st, 0, d_o, d_d, stream)
cnt = torch.zeros((n, 2), dtype=torch.int32, device=dev)
scene.trace_counts_device(d_o, d_d, n, cnt, stream)
torch.cuda.synchronize()
c = cnt.cpu().numpy().astype(np.int64)
print(os.environ.get("RTGPU_LIB", "default"), "boxes/ray %.2f tris/ray %.2f" % (c[:, 0].mean(), c[:, 1].mean()), "nodes", scene.info().n_nodes)
