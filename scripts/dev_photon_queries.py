"""Developer measurement: the gather over the primary hits of the configs[2] frame -- how many queries are cheap."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch  # noqa: F401
import oracle
from raytracer_amd import abi, api, scenes

sc, cam_m, yfov = scenes.interior_scene(scenes.C3_DETAIL)
W, H = 480, 270
cam = api.camera_from_transform(scenes.x_fov_degrees(yfov, W / H), cam_m)
o, d = oracle.generate_primary(cam, abi.make_settings(W, H, geometry_aa=1, material_aa=1, jitter_mode=abi.JITTER_CENTER))
with api.Scene(sc) as g:
    n = g.build_photon_map(abi.PM_FULL, 0, 1_000_000, 1_000_000, seed=1)
    h = g.trace_rays(o, d)
    hit = h["inst"] != abi.INVALID_ID
    q = (o + d * h["t"][:, None])[hit].astype(np.float32)
    qd = (-d[hit]).astype(np.float32)
    print("photons", n, "queries", len(q), flush=True)
    for lo in range(0, len(q), 16384):
        t0 = time.perf_counter()
        out = g.gather_photons(q[lo:lo + 16384], qd[lo:lo + 16384])
        dt = time.perf_counter() - t0
        print("chunk %6d: %.3f s, count<20: %d, max_dist p50 %.4f p99 %.4f max %.4f" % (lo, dt, (out["count"] < 20).sum(), np.median(out["max_dist"]),
              np.percentile(out["max_dist"], 99), out["max_dist"][out["count"] == 20].max()), flush=True)
    # what a hard query looks like
    out = g.gather_photons(q[:16384], qd[:16384])
    hard = np.argmax(out["max_dist"])
    ph = g.photons()
    dist = np.linalg.norm(ph["pos"] - q[hard], axis=1)
    near = dist < out["max_dist"][hard]
    dots = ph["normal"][near] @ qd[hard]
    print("hard query", hard, "point", q[hard], "dir", qd[hard], "radius", out["max_dist"][hard], "photons inside", near.sum(), "passing", (dots >= 0).sum())
    nn = ph["normal"][near]
    u, c = np.unique(np.round(nn, 2), axis=0, return_counts=True)
    order = np.argsort(-c)[:8]
    for k in order:
        print("   normal", u[k], "count", c[k], "dot", float(u[k] @ qd[hard]))
