"""Developer probe: share of node steps that enter neither child (mostly stale stack entries) in
the single-ray traversal.  Needs the dev library and RTGPU_COUNT_COOPERATIVE=0."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from raytracer_amd import scenes, abi, api
dev = torch.device("cuda", 0); stream = torch.cuda.current_stream()
def probe(g, o, d, label):
    n = len(o)
    d_o = torch.from_numpy(np.ascontiguousarray(o)).to(dev); d_d = torch.from_numpy(np.ascontiguousarray(d)).to(dev)
    cnt = torch.zeros((n, 2), dtype=torch.int32, device=dev)
    g.trace_counts_device(d_o, d_d, n, cnt, stream); torch.cuda.synchronize()
    c = cnt.cpu().numpy().view(np.uint32).astype(np.int64)
    steps = c[:, 0] / 2; dead = c[:, 1] >> 16; tris = c[:, 1] & 0xFFFF
    print(f"{label}: node steps/ray {steps.mean():.2f} of which enter no child {dead.mean():.2f} ({100*dead.sum()/steps.sum():.1f} %), tris/ray {tris.mean():.2f}")
W, H = 1920, 1080
sc, cam_m, yfov = scenes.blob_scene(8)
g = api.Scene(sc)
cam = api.camera_from_transform(scenes.x_fov_degrees(yfov, W / H), cam_m)
st = abi.make_settings(W, H, geometry_aa=1, material_aa=1, jitter_mode=abi.JITTER_CENTER)
d_o = torch.empty((W*H, 3), dtype=torch.float32, device=dev); d_d = torch.empty((W*H, 3), dtype=torch.float32, device=dev)
api.generate_primary_device(cam, st, 0, d_o, d_d, stream); torch.cuda.synchronize()
o = d_o.cpu().numpy(); d = d_d.cpu().numpy()
probe(g, o, d, "blob primary")
hits = g.trace_rays(o, d); m = hits["inst"] != abi.INVALID_ID
probe(g, o[m], d[m], "blob primary, hit rays only")
hp = (o[m] + d[m] * hits["t"][m][:, None]).astype(np.float32)
rng = np.random.RandomState(1)
sd = rng.normal(size=hp.shape); sd /= np.linalg.norm(sd, axis=1, keepdims=True); nn = hp / np.linalg.norm(hp, axis=1, keepdims=True)
fl = (sd * nn).sum(1) < 0; sd[fl] *= -1
probe(g, (hp + nn * 1e-3).astype(np.float32), sd.astype(np.float32), "blob bounce")
g.close()
sc, cam_m, yfov = scenes.interior_scene(scenes.C3_DETAIL)
g = api.Scene(sc)
ro, rd = scenes.random_rays(1_000_000, (1, 1, 1), (23, 8, 9), seed=3)
probe(g, ro, rd, "interior random")
