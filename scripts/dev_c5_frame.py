"""Developer probe for configs[4]: the path-traced 1080p 4x4 frame of the flattened 49.8 M-triangle cluster (frame time, trace
and shade stage of a profiling render), for whatever build / kernel knobs the environment sets."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from raytracer_amd import scenes, abi, api

dev = torch.device("cuda", 0)
W, H = 1920, 1080
copies = int(sys.argv[1]) if len(sys.argv) > 1 else 38
G = int(sys.argv[2]) if len(sys.argv) > 2 else 4
M = int(sys.argv[3]) if len(sys.argv) > 3 else 4
sc, cam_m, yfov = scenes.cluster_scene(copies, 8, flatten=True)
g = api.Scene(sc)
info = g.info()
cam = api.camera_from_transform(scenes.x_fov_degrees(yfov, W / H), cam_m)
fst = abi.make_settings(W, H, geometry_aa=G, material_aa=M, max_path_length=9, seed=1)
fb = torch.zeros((H, W, 3), dtype=torch.float32, device=dev)
r = api.Renderer()
r.render_device(g, cam, fst, fb)
s = r.render_device(g, cam, fst, fb)
p = r.render_profile_device(g, cam, fst, fb)
p = r.render_profile_device(g, cam, fst, fb)
knobs = " ".join("%s=%s" % (k, v) for k, v in sorted(os.environ.items()) if k.startswith("RTGPU_") and k != "RTGPU_LIB")
print("%dx%d spp | %s | nodes %d depth %d bytes %.2f GB build %.1f s | frame %.2f ms trace %.2f ms (%.0f Mrays/s) shade %.2f ms | img %.6f" % (
    G, M, knobs, info.n_nodes, info.max_depth, info.device_bytes / 1e9, info.build_ms / 1e3, s.device_ms, p.trace_ms,
    (s.extension_rays + s.shadow_rays) / p.trace_ms / 1e3, p.shade_ms, float(fb[torch.isfinite(fb)].mean())), flush=True)
