"""Profiling target for configs[2] / configs[3]: path-traced frames of the interior scene, one batch in flight (run with
RTGPU_RENDER_LANES=1).  Argument "1080p": two frames 1920x1080, 8x8 spp (configs[2]); "4k": one frame 3840x2160, 16x16 spp
(configs[3] on one GPU).  Path length 9, seed 1 -- the frames bench.py's extras render."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from raytracer_amd import abi, api, scenes

which = sys.argv[1] if len(sys.argv) > 1 else "1080p"
W, H, G, M, frames = (3840, 2160, 16, 16, 1) if which == "4k" else (1920, 1080, 8, 8, 2)
dev = torch.device("cuda", 0)
sc, cam_m, yfov = scenes.interior_scene(scenes.C3_DETAIL)
cam = api.camera_from_transform(scenes.x_fov_degrees(yfov, W / H), cam_m)
scene = api.Scene(sc)
st = abi.make_settings(W, H, geometry_aa=G, material_aa=M, max_path_length=9, seed=1)
fb = torch.zeros((H, W, 3), dtype=torch.float32, device=dev)
r = api.Renderer()
for _ in range(frames):
    s = r.render_device(scene, cam, st, fb)
torch.cuda.synchronize()
print("frame %.1f ms, %d rays" % (s.device_ms, s.primary_rays + s.extension_rays + s.shadow_rays))
