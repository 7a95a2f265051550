"""Developer measurement: does the order of the queries matter to the photon gather?  1 M queries near stored photons,
as picked (random), and sorted along a Morton curve."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch  # noqa: F401  (HIP runtime first)
from raytracer_amd import abi, api, scenes

sc, cam_m, yfov = scenes.interior_scene(scenes.C3_DETAIL)
with api.Scene(sc) as g:
    g.build_photon_map(abi.PM_FULL, 0, 1_000_000, 1_000_000, seed=1)
    ph = g.photons()
    rng = np.random.RandomState(0)
    pick = rng.randint(0, len(ph), 1_000_000)
    q = (ph["pos"][pick] + rng.randn(1_000_000, 3) * 0.01).astype(np.float32)
    d = ph["normal"][pick]
    lo, hi = q.min(0), q.max(0)
    cell = np.clip(((q - lo) / (hi - lo) * 1023).astype(np.uint64), 0, 1023)
    code = np.zeros(len(q), np.uint64)
    for bit in range(10):
        for k in range(3):
            code |= ((cell[:, k] >> np.uint64(bit)) & np.uint64(1)) << np.uint64(3 * bit + k)
    order = np.argsort(code, kind="stable")
    for name, (qq, dd) in (("as picked", (q, d)), ("Morton order", (q[order], d[order])), ("as picked", (q, d))):
        g.gather_photons(qq[:1000], dd[:1000])
        t0 = time.perf_counter()
        out = g.gather_photons(np.ascontiguousarray(qq), np.ascontiguousarray(dd))
        dt = time.perf_counter() - t0
        print("%-14s %.3f s = %.1f M queries/s" % (name, dt, len(qq) / dt / 1e6), flush=True)
