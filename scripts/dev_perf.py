"""Developer perf probe: coherent / incoherent closest-hit and a C3-style render."""
import sys, time, os, argparse
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from raytracer_amd import scenes, abi, api

ap = argparse.ArgumentParser()
ap.add_argument("--what", default="trace,render")
ap.add_argument("--subdiv", type=int, default=8)
ap.add_argument("--detail", type=float, default=1.78)
ap.add_argument("--spp", type=int, default=8)
ap.add_argument("--w", type=int, default=1920)
ap.add_argument("--h", type=int, default=1080)
args = ap.parse_args()
dev = torch.device("cuda", 0)
stream = torch.cuda.current_stream()

def time_trace(g, o, d, reps=10, any_tmax=None):
    n = o.shape[0]
    d_o = torch.from_numpy(o).to(dev); d_d = torch.from_numpy(d).to(dev)
    hits = torch.empty((n, 5), dtype=torch.int32, device=dev)
    if any_tmax is not None:
        d_t = torch.from_numpy(any_tmax).to(dev); occ = torch.empty(n, dtype=torch.uint8, device=dev)
    f = (lambda: g.test_visibility_device(d_o, d_d, d_t, n, occ, stream)) if any_tmax is not None else (lambda: g.trace_rays_device(d_o, d_d, n, hits, stream))
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record(stream)
    for _ in range(reps): f()
    e1.record(stream); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    cnt = torch.zeros((n, 2), dtype=torch.int32, device=dev)
    g.trace_counts_device(d_o, d_d, n, cnt, stream); torch.cuda.synchronize()
    c = cnt.double().mean(0).tolist()
    return ms, n / ms / 1e3, c

if "trace" in args.what:
    sc, cam_m, yfov = scenes.blob_scene(args.subdiv)
    g = api.Scene(sc); info = g.info()
    print(f"blob: tris={info.n_triangles} nodes={info.n_nodes} depth={info.max_depth} build={info.build_ms:.0f}ms bytes={info.device_bytes/1e6:.0f}MB")
    cam = api.camera_from_transform(scenes.x_fov_degrees(yfov, args.w / args.h), cam_m)
    st = abi.make_settings(args.w, args.h, geometry_aa=1, material_aa=1, jitter_mode=abi.JITTER_CENTER)
    n = args.w * args.h
    d_o = torch.empty((n, 3), dtype=torch.float32, device=dev); d_d = torch.empty((n, 3), dtype=torch.float32, device=dev)
    api.generate_primary_device(cam, st, 0, d_o, d_d, stream); torch.cuda.synchronize()
    o = d_o.cpu().numpy(); d = d_d.cpu().numpy()
    ms, mr, c = time_trace(g, o, d)
    print(f"  primary coherent: {ms:.3f} ms {mr:.0f} Mrays/s boxes/ray={c[0]:.1f} tris/ray={c[1]:.1f}")
    # incoherent: diffuse bounce rays from the primary hit points
    hits = g.trace_rays(o, d)
    m = hits["inst"] != abi.INVALID_ID
    hp = (o[m] + d[m] * hits["t"][m][:, None]).astype(np.float32)
    rng = np.random.RandomState(1)
    dirs = rng.normal(size=hp.shape); dirs /= np.linalg.norm(dirs, axis=1, keepdims=True)
    nrm = hp / np.linalg.norm(hp, axis=1, keepdims=True)
    flip = (dirs * nrm).sum(1) < 0; dirs[flip] *= -1
    so = (hp + nrm * 1e-3).astype(np.float32); sd = dirs.astype(np.float32)
    reps = int(np.ceil(2_000_000 / len(so)))
    so = np.tile(so, (reps, 1)); 
    sd = rng.normal(size=so.shape); sd /= np.linalg.norm(sd, axis=1, keepdims=True); nn = so / np.linalg.norm(so, axis=1, keepdims=True)
    fl = (sd * nn).sum(1) < 0; sd[fl] *= -1; sd = sd.astype(np.float32)
    ms, mr, c = time_trace(g, so, sd)
    print(f"  bounce incoherent (surface->hemisphere): n={len(so)} {ms:.3f} ms {mr:.0f} Mrays/s boxes/ray={c[0]:.1f} tris/ray={c[1]:.1f}")
    ro, rd = scenes.random_rays(2_000_000, (-1.6, -1.6, -1.6), (1.6, 1.6, 1.6), seed=2)
    ms, mr, c = time_trace(g, ro, rd)
    print(f"  random incoherent (box->sphere): {ms:.3f} ms {mr:.0f} Mrays/s boxes/ray={c[0]:.1f} tris/ray={c[1]:.1f}")
    ms, mr, c = time_trace(g, ro, rd, any_tmax=np.full(len(ro), 0.5, np.float32))
    print(f"  random any-hit tmax=0.5: {ms:.3f} ms {mr:.0f} Mrays/s")
    g.close()

if "render" in args.what:
    sc, cam_m, yfov = scenes.interior_scene(detail=args.detail)
    t0 = time.time(); g = api.Scene(sc); info = g.info()
    print(f"interior: tris={info.n_triangles} nodes={info.n_nodes} depth={info.max_depth} build={info.build_ms:.0f}ms")
    cam = api.camera_from_transform(scenes.x_fov_degrees(yfov, args.w / args.h), cam_m)
    st = abi.make_settings(args.w, args.h, geometry_aa=args.spp, material_aa=args.spp, max_path_length=9, seed=1)
    fb = torch.zeros((args.h, args.w, 3), dtype=torch.float32, device=dev)
    r = api.Renderer()
    for rep in range(2):
        t0 = time.time(); stats = r.render_device(g, cam, st, fb, stream); t1 = time.time()
        rays = stats.primary_rays + stats.extension_rays + stats.shadow_rays
        print(f"  render {args.w}x{args.h} {args.spp}x{args.spp}: wall {t1-t0:.3f}s dev {stats.device_ms:.1f} ms rays p/e/s={stats.primary_rays}/{stats.extension_rays}/{stats.shadow_rays} -> {rays/stats.device_ms/1e3:.0f} Mrays/s mean={fb[torch.isfinite(fb)].mean().item():.4f} nonfinite={int((~torch.isfinite(fb)).sum().item())}")
    g.close()
