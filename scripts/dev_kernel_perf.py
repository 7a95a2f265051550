"""Developer probe: closest-hit rate on configs[1] primary and bounce rays, any-hit rate on shadow-like rays, and the trace
stage of a 1080p 8x8 frame of the blob-in-a-room scene (profiling render), for the library named by RTGPU_LIB."""
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
hits = torch.empty((n, 5), dtype=torch.int32, device=dev)


def timed(f, reps=20):
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(stream)
    for _ in range(reps):
        f()
    e1.record(stream)
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


ms = timed(lambda: scene.trace_rays_device(d_o, d_d, n, hits, stream))
res = ["primary %.0f" % (n / ms / 1e3)]
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
ms = timed(lambda: scene.trace_rays_device(b_o, b_d, n, hits, stream))
res.append("bounce %.0f" % (n / ms / 1e3))
chk = int(hits.cpu().numpy().view(np.uint32)[:, 1].astype(np.uint64).sum() % 1000003)
tmax = torch.full((n,), 2.5, dtype=torch.float32, device=dev)
occ = torch.empty(n, dtype=torch.uint8, device=dev)
ms = timed(lambda: scene.test_visibility_device(b_o, b_d, tmax, n, occ, stream))
res.append("any %.0f" % (n / ms / 1e3))
scene.close()
if "--no-frame" not in sys.argv:
    sc, cam_m, yfov = scenes.blob_room_scene(8)
    cam = api.camera_from_transform(scenes.x_fov_degrees(yfov, W / H), cam_m)
    scene = api.Scene(sc)
    st = abi.make_settings(W, H, geometry_aa=8, material_aa=8, max_path_length=9, seed=1)
    fb = torch.zeros((H, W, 3), dtype=torch.float32, device=dev)
    r = api.Renderer()
    r.render_device(scene, cam, st, fb)
    s = r.render_device(scene, cam, st, fb)
    p = r.render_profile_device(scene, cam, st, fb)
    rays = s.extension_rays + s.shadow_rays
    res.append("frame %.1f ms trace %.1f ms (%.0f Mrays/s) shade %.1f ms primary %.2f ms resolve %.2f ms" % (s.device_ms, p.trace_ms, rays / p.trace_ms / 1e3, p.shade_ms, p.primary_ms, p.resolve_ms))
    res.append("img %.6f bits %d" % (float(fb[torch.isfinite(fb)].mean()), int(fb.view(torch.int32).to(torch.int64).sum().item() % 1000000007)))
print(os.environ.get("RTGPU_LIB", "default"), "|", " | ".join(res), "| chk", chk)
