"""Developer measurement: rtgpu_scene_build_photon_map on the interior scene (RTGPU_PM_STATS=1 prints the phases)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: F401  (HIP runtime first)
from raytracer_amd import abi, api, scenes

sc, cam_m, yfov = scenes.interior_scene(scenes.C3_DETAIL)
with api.Scene(sc) as g:
    g.build_photon_map(abi.PM_FULL, 0, 20000, 20000, seed=1)
    t0 = time.perf_counter()
    n = g.build_photon_map(abi.PM_FULL, 0, 1_000_000, 1_000_000, seed=1)
    print("build_photon_map: %d photons in %.2f s" % (n, time.perf_counter() - t0), flush=True)
