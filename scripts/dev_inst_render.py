"""Developer measurement: a frame of a scene of mesh instances (the reference's default two-level mode) beside the
same scene flattened into one mesh."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch  # noqa: F401  (HIP runtime first)
from raytracer_amd import abi, api, scenes

W, H = 1920, 1080
for flatten in (False, True):
    sc, cam_m, yfov = scenes.cluster_scene(12, 6, flatten=flatten)
    if not sc.point_lights:
        sc.point_lights = [((0.0, 6.0, 4.0), (1.0, 1.0, 1.0), 400.0)]
    sc.env = ((0.4, 0.5, 0.7), 0.6)
    cam = api.camera_from_transform(scenes.x_fov_degrees(yfov, W / H), cam_m)
    st = abi.make_settings(W, H, geometry_aa=4, material_aa=4, max_path_length=6, seed=1)
    with api.Scene(sc) as g:
        api.Renderer().render(g, cam, abi.make_settings(W, H, geometry_aa=1, material_aa=1, seed=1))
        for _ in range(2):
            img, stats = api.Renderer().render(g, cam, st)
        rays = stats.primary_rays + stats.extension_rays + stats.shadow_rays
        print("%s: %d triangles, frame %.1f ms, %d rays, %.0f Mrays/s, mean %.4f" % ("flattened" if flatten else "instanced", sc.n_triangles() if flatten else -1,
              stats.device_ms, rays, rays / stats.device_ms / 1e3, img.mean()), flush=True)
