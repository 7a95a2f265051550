"""Developer check: non-finite pixels of the C3 render, GPU vs oracle on the affected pixels."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import oracle
from raytracer_amd import scenes, abi, api

W, H, G, M = 1920, 1080, 8, 8
detail = float(sys.argv[1]) if len(sys.argv) > 1 else scenes.C3_DETAIL
sc, cam_m, yfov = scenes.interior_scene(detail)
cam = api.camera_from_transform(scenes.x_fov_degrees(yfov, W / H), cam_m)
st = abi.make_settings(W, H, geometry_aa=G, material_aa=M, max_path_length=9, seed=1)
g = api.Scene(sc)
img, stats = api.Renderer().render(g, cam, st)
bad = np.argwhere(~np.isfinite(img).all(axis=2))
print("non-finite pixels:", len(bad), bad[:10].tolist())
S = oracle.OracleScene(sc)
for (y, x) in bad[:4]:
    st2 = abi.make_settings(W, H, geometry_aa=G, material_aa=M, max_path_length=9, seed=1, tile=(int(x), int(y), int(x) + 1, int(y) + 1))
    ref, _ = S.render(cam, st2)
    img2, _ = api.Renderer().render(g, cam, st2)
    print("pixel", x, y, "gpu full", img[y, x], "gpu tile", img2[y, x], "oracle", ref[y, x])
