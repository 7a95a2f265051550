"""Developer measurement: the interior scene as one mesh and cut into many mesh instances (what a file with many
objects gives in the reference's default two-level mode)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch  # noqa: F401  (HIP runtime first)
from raytracer_amd import abi, api, scenes
from raytracer_amd.abi import InstanceData, MeshData

W, H = 1920, 1080
sc, cam_m, yfov = scenes.interior_scene(scenes.C3_DETAIL)
cam = api.camera_from_transform(scenes.x_fov_degrees(yfov, W / H), cam_m)
st = abi.make_settings(W, H, geometry_aa=4, material_aa=4, max_path_length=9, seed=1)


def cut(sc, cells):
    m = sc.meshes[0]
    cen = m.vertices[m.indices].mean(1)
    lo, hi = cen.min(0), cen.max(0)
    key = np.clip(((cen - lo) / (hi - lo + 1e-9) * cells).astype(int), 0, cells - 1)
    key = (key[:, 0] * cells + key[:, 1]) * cells + key[:, 2]
    meshes, insts = [], []
    for k in np.unique(key):
        sel = np.nonzero(key == k)[0]
        used, inv = np.unique(m.indices[sel].reshape(-1), return_inverse=True)
        meshes.append(MeshData(m.vertices[used], m.normals[used], inv.reshape(-1, 3).astype(np.uint32),
                               None if m.tri_material is None else m.tri_material[sel], None if m.texcoords is None else m.texcoords[used]))
        insts.append(InstanceData(len(meshes) - 1, sc.instances[0].material, np.eye(4), np.eye(4)))
    sc.meshes, sc.instances = meshes, insts
    return sc


for cells in (0, 3, 6):
    s2, _, _ = scenes.interior_scene(scenes.C3_DETAIL)
    if cells:
        s2 = cut(s2, cells)
    with api.Scene(s2) as g:
        api.Renderer().render(g, cam, abi.make_settings(W, H, geometry_aa=1, material_aa=1, seed=1))
        for _ in range(2):
            img, stats = api.Renderer().render(g, cam, st)
        rays = stats.primary_rays + stats.extension_rays + stats.shadow_rays
        print("%4d instances: frame %.1f ms, %d rays, %.0f Mrays/s, mean %.4f" % (len(s2.instances), stats.device_ms, rays, rays / stats.device_ms / 1e3, img.mean()), flush=True)
