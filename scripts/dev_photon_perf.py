"""Developer measurement: photon tracing, tree build and gather rates on the interior scene (configs[2] geometry)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch  # noqa: F401  (HIP runtime first)
from raytracer_amd import abi, api, scenes

sc, cam_m, yfov = scenes.interior_scene(scenes.C3_DETAIL)
W, H = 1920, 1080
cam = api.camera_from_transform(scenes.x_fov_degrees(yfov, W / H), cam_m)
with api.Scene(sc) as g:
    for rays in (250_000, 1_000_000):
        t0 = time.perf_counter()
        n = g.build_photon_map(abi.PM_FULL, 0, rays, rays, seed=1)
        dt = time.perf_counter() - t0
        print("build_photon_map: %d rays per light (%d lights) -> %d photons in %.2f s" % (rays, len(sc.point_lights) + len(sc.area_lights), n, dt), flush=True)
    ph = g.photons()
    rng = np.random.RandomState(0)
    pick = rng.randint(0, len(ph), 1_000_000)
    q = (ph["pos"][pick] + rng.randn(1_000_000, 3) * 0.01).astype(np.float32)
    d = ph["normal"][pick]   # a direction the local photons pass the filter with (one that rejects them all makes the search visit the whole map)
    g.gather_photons(q[:1000], d[:1000])
    t0 = time.perf_counter()
    out = g.gather_photons(q, d)
    dt = time.perf_counter() - t0
    print("gather (host arrays in/out): %d queries over %d photons in %.3f s = %.1f M queries/s, mean count %.1f" % (len(q), len(ph), dt, len(q) / dt / 1e6, out["count"].mean()), flush=True)
    for depth in (0, 1):
        g.set_photons(ph, abi.PM_FULL, depth)
        st = abi.make_settings(W, H, geometry_aa=4, material_aa=4, max_path_length=9, seed=1)
        api.Renderer().render(g, cam, abi.make_settings(W, H, geometry_aa=1, material_aa=1, seed=1))
        t0 = time.perf_counter()
        img, stats = api.Renderer().render(g, cam, st)
        dt = time.perf_counter() - t0
        print("render 1080p 4x4 spp, photon map depth %d: %.1f ms device %.1f ms, mean %.4f" % (depth, dt * 1e3, stats.device_ms, img.mean()), flush=True)
