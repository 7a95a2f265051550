"""Developer check: the work-stealing kernels give identical results run to run and equal to the per-lane kernel."""
import sys, os, subprocess
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from raytracer_amd import scenes, abi, api
mode = sys.argv[1] if len(sys.argv) > 1 else "main"
sc, cam_m, yfov = scenes.interior_scene(detail=1.0)
g = api.Scene(sc)
o, d = scenes.random_rays(3_000_000, (0.5, 0.5, 0.5), (23, 8, 9), seed=4)
h = g.trace_rays(o, d)
tm = np.where(np.isfinite(h["t"]), h["t"] * 0.999, 5.0).astype(np.float32)
occ = g.test_visibility(o, d, tm)
if mode == "dump":
    np.save("/tmp/det_h.npy", h); np.save("/tmp/det_occ.npy", occ)
    sys.exit(0)
for rep in range(3):
    h2 = g.trace_rays(o, d); occ2 = g.test_visibility(o, d, tm)
    print("rep", rep, "hits identical:", h.tobytes() == h2.tobytes(), "occ identical:", np.array_equal(occ, occ2))
env = dict(os.environ, RTGPU_TRACE_VARIANT="-1")
subprocess.check_call([sys.executable, __file__, "dump"], env=env)
hp = np.load("/tmp/det_h.npy"); op = np.load("/tmp/det_occ.npy")
same = h.tobytes() == hp.tobytes()
print("steal == per-lane kernel:", same, "occ:", np.array_equal(occ, op))
if not same:
    for f in h.dtype.names:
        bad = np.nonzero(h[f] != hp[f])[0]
        nanboth = np.isnan(h[f][bad].astype(np.float64)) & np.isnan(hp[f][bad].astype(np.float64)) if f in ("t", "beta", "gamma") else np.zeros(len(bad), bool)
        print(" field", f, "diffs", len(bad), "both-nan", int(nanboth.sum()), h[bad[:3]], hp[bad[:3]])
