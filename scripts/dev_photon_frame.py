"""Developer measurement: one photon-mapped frame (mode full, depth 0) of the interior scene, for kernel profiles."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch  # noqa: F401  (HIP runtime first)
from raytracer_amd import abi, api, scenes

sc, cam_m, yfov = scenes.interior_scene(scenes.C3_DETAIL)
W, H = 1920, 1080
cam = api.camera_from_transform(scenes.x_fov_degrees(yfov, W / H), cam_m)
depth = int(sys.argv[1]) if len(sys.argv) > 1 else 0
with api.Scene(sc) as g:
    n = g.build_photon_map(abi.PM_FULL, depth, 1_000_000, 1_000_000, seed=1)
    st = abi.make_settings(W, H, geometry_aa=4, material_aa=4, max_path_length=9, seed=1)
    api.Renderer().render(g, cam, abi.make_settings(W, H, geometry_aa=1, material_aa=1, seed=1))
    t0 = time.perf_counter()
    img, stats = api.Renderer().render(g, cam, st)
    dt = time.perf_counter() - t0
    print("photons %d, render 1080p 4x4 spp depth %d: %.1f ms device %.1f ms, mean %.4f" % (n, depth, dt * 1e3, stats.device_ms, img.mean()), flush=True)
