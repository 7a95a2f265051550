"""Developer probe for config 5: build and trace a flattened ~50 M-triangle cluster and compare
its hits with the instanced version of the same scene."""
import sys, os, time, argparse
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from raytracer_amd import scenes, abi, api

ap = argparse.ArgumentParser()
ap.add_argument("--copies", type=int, default=38)
ap.add_argument("--subdiv", type=int, default=8)
args = ap.parse_args()
dev = torch.device("cuda", 0); stream = torch.cuda.current_stream()
W, H = 1920, 1080
t0 = time.time(); sc, cam_m, yfov = scenes.cluster_scene(args.copies, args.subdiv, flatten=True); t1 = time.time()
print(f"generated {sc.n_triangles()} triangles in {t1-t0:.1f}s", flush=True)
g = api.Scene(sc); t2 = time.time(); info = g.info()
print(f"scene create {t2-t1:.1f}s (tree build {info.build_ms/1e3:.1f}s) nodes={info.n_nodes} depth={info.max_depth} device bytes={info.device_bytes/1e9:.2f} GB", flush=True)
cam = api.camera_from_transform(scenes.x_fov_degrees(yfov, W / H), cam_m)
st = abi.make_settings(W, H, geometry_aa=1, material_aa=1, jitter_mode=abi.JITTER_CENTER)
n = W * H
d_o = torch.empty((n, 3), dtype=torch.float32, device=dev); d_d = torch.empty((n, 3), dtype=torch.float32, device=dev)
api.generate_primary_device(cam, st, 0, d_o, d_d, stream)
hits = torch.empty((n, 5), dtype=torch.int32, device=dev)
for _ in range(3): g.trace_rays_device(d_o, d_d, n, hits, stream)
torch.cuda.synchronize()
e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
e0.record(stream)
for _ in range(10): g.trace_rays_device(d_o, d_d, n, hits, stream)
e1.record(stream); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 10
hf = hits.cpu().numpy().view(np.uint32)
print(f"flattened primary: {ms:.3f} ms {n/ms/1e3:.0f} Mrays/s hit fraction {(hf[:,2]!=0xFFFFFFFF).mean():.3f}", flush=True)
# the path-traced frame of configs[4]: 1080p, 4x4 spp, path length 9
fst = abi.make_settings(W, H, geometry_aa=4, material_aa=4, max_path_length=9, seed=1)
fb = torch.zeros((H, W, 3), dtype=torch.float32, device=dev)
r = api.Renderer()
r.render_device(g, cam, fst, fb)
s = r.render_device(g, cam, fst, fb)
if hasattr(r, "render_profile_device"):
    p = r.render_profile_device(g, cam, fst, fb)
    print(f"frame 4x4: {s.device_ms:.1f} ms, {(s.primary_rays + s.extension_rays + s.shadow_rays) / s.device_ms / 1e3:.0f} Mrays/s; trace stage {p.trace_ms:.1f} ms = {(s.extension_rays + s.shadow_rays) / p.trace_ms / 1e3:.0f} Mrays/s", flush=True)
else:
    print(f"frame 4x4: {s.device_ms:.1f} ms, {(s.primary_rays + s.extension_rays + s.shadow_rays) / s.device_ms / 1e3:.0f} Mrays/s", flush=True)
del fb
g.close()
sc2, _, _ = scenes.cluster_scene(args.copies, args.subdiv, flatten=False)
g2 = api.Scene(sc2)
hits2 = torch.empty((n, 5), dtype=torch.int32, device=dev)
for _ in range(3): g2.trace_rays_device(d_o, d_d, n, hits2, stream)
torch.cuda.synchronize()
e0.record(stream)
for _ in range(10): g2.trace_rays_device(d_o, d_d, n, hits2, stream)
e1.record(stream); torch.cuda.synchronize()
ms2 = e0.elapsed_time(e1) / 10
hi = hits2.cpu().numpy().view(np.uint32)
print(f"instanced primary: {ms2:.3f} ms {n/ms2/1e3:.0f} Mrays/s", flush=True)
nt = sc2.meshes[0].indices.shape[0]
hit_f = hf[:, 2] != 0xFFFFFFFF; hit_i = hi[:, 2] != 0xFFFFFFFF
same_hit = hit_f == hit_i
both = hit_f & hit_i
same_id = hf[both, 1].astype(np.int64) == hi[both, 2].astype(np.int64) * nt + hi[both, 1]
tf = hf[:, 0].view(np.float32); ti = hi[:, 0].view(np.float32)
rel = np.abs(tf[both] - ti[both]) / np.maximum(np.abs(ti[both]), 1e-6)
print(f"hit/miss agreement {same_hit.mean():.6f}; same triangle {same_id.mean():.6f}; max rel t diff (same triangle) {rel[same_id].max():.3g}")
