"""Developer check: caustics-mode parity on the glass scene with and without photons stacked on one spot."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import oracle
from raytracer_amd import abi, api, scenes

sc, cam_m, yfov = scenes.glass_scene()
S = oracle.OracleScene(sc)
S.build_photon_map(abi.PM_CAUSTICS, 0, 25000, 25000, seed=5)
ph_all = S.photons()
key = ph_all["pos"].view(np.uint32).reshape(-1, 3)
_, first = np.unique(key, axis=0, return_index=True)
cam = api.camera_from_transform(scenes.x_fov_degrees(yfov, 64 / 48), cam_m)
st = abi.make_settings(64, 48, geometry_aa=4, material_aa=8, seed=11)
for name, ph in (("all", ph_all), ("unique positions", ph_all[np.sort(first)])):
    S.set_photons(ph, abi.PM_CAUSTICS, 0)
    ref, _ = S.render(cam, st)
    with api.Scene(sc) as g:
        g.set_photons(ph, abi.PM_CAUSTICS, 0)
        img, _ = api.Renderer().render(g, cam, st)
    close = np.isclose(img, ref, rtol=2e-3, atol=1e-5).all(axis=2)
    print(name, len(ph), "close", close.mean(), "rel rmse", np.sqrt(np.mean((img - ref) ** 2)) / ref.mean())
