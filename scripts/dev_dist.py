"""Developer probe: distribution of traversal work per ray / per wave on the C2 primary rays."""
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
api.generate_primary_device(cam, st, 0, d_o, d_d, stream)
cnt = torch.zeros((n, 2), dtype=torch.int32, device=dev)
g.trace_counts_device(d_o, d_d, n, cnt, stream); torch.cuda.synchronize()
c = cnt.cpu().numpy().astype(np.int64)
steps = c[:, 0] // 2 + c[:, 1]
print("per ray steps: mean %.1f p50 %d p90 %d p99 %d p99.9 %d max %d" % (steps.mean(), *np.percentile(steps, [50, 90, 99, 99.9]), steps.max()))
w = steps.reshape(-1, 64)
wmax = w.max(axis=1); wmean = w.mean(axis=1)
print("per wave max-steps: mean %.1f p50 %d p90 %d p99 %d max %d ; SIMT efficiency = %.3f" % (wmax.mean(), *np.percentile(wmax, [50, 90, 99]), wmax.max(), wmean.sum() / wmax.sum()))
# 8x8 tiles instead of 64x1 spans
img = steps.reshape(H, W)
t = img[:H // 8 * 8, :].reshape(H // 8, 8, W // 8, 8).transpose(0, 2, 1, 3).reshape(-1, 64)
print("8x8 tiles: per wave max-steps mean %.1f ; SIMT efficiency = %.3f" % (t.max(axis=1).mean(), t.mean(axis=1).sum() / t.max(axis=1).sum()))
