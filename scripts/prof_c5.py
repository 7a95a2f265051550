"""Profiling target for configs[4]: two path-traced frames (1920x1080, 4x4 spp, path length 9) of the 49.8 M-triangle
cluster, one batch in flight (run with RTGPU_RENDER_LANES=1)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from raytracer_amd import abi, api, scenes

dev = torch.device("cuda", 0)
W, H = 1920, 1080
sc, cam_m, yfov = scenes.cluster_scene(38, 8, flatten=True)
cam = api.camera_from_transform(scenes.x_fov_degrees(yfov, W / H), cam_m)
scene = api.Scene(sc)
st = abi.make_settings(W, H, geometry_aa=4, material_aa=4, max_path_length=9, seed=1)
fb = torch.zeros((H, W, 3), dtype=torch.float32, device=dev)
r = api.Renderer()
for _ in range(2):
    s = r.render_device(scene, cam, st, fb)
torch.cuda.synchronize()
print("frame %.1f ms, %d rays" % (s.device_ms, s.primary_rays + s.extension_rays + s.shadow_rays))
