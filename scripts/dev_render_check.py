"""Developer check: GPU wavefront render vs the CPU oracle render (same counter-based RNG streams)."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import oracle
from raytracer_amd import scenes, abi, api

def psnr(a, b):
    mse = float(np.mean((a - b) ** 2))
    peak = float(max(a.max(), b.max(), 1e-6))
    return 10 * np.log10(peak * peak / max(mse, 1e-30)), mse

def check(name, sc, cam_m, yfov, W, H, G, M, seed=1, maxlen=10):
    cam = api.camera_from_transform(scenes.x_fov_degrees(yfov, W / H), cam_m)
    st = abi.make_settings(W, H, geometry_aa=G, material_aa=M, max_path_length=maxlen, seed=seed)
    S = oracle.OracleScene(sc)
    t0 = time.time(); ref, rstats = S.render(cam, st); t1 = time.time()
    g = api.Scene(sc)
    t2 = time.time(); img, stats = api.Renderer().render(g, cam, st); t3 = time.time()
    p, mse = psnr(ref, img)
    rel = np.abs(ref - img) / np.maximum(np.abs(ref), 1e-3)
    nbad = int((rel.max(axis=2) > 1e-3).sum())
    print(f"{name}: {W}x{H} G={G} M={M} mean ref={ref.mean():.5f} gpu={img.mean():.5f} PSNR={p:.1f} dB rmse={np.sqrt(mse):.3g} pixels>1e-3 rel: {nbad}/{W*H}"
          f" rays ref p/e/s={rstats.primary_rays}/{rstats.extension_rays}/{rstats.shadow_rays} gpu={stats.primary_rays}/{stats.extension_rays}/{stats.shadow_rays}"
          f" oracle {t1-t0:.2f}s gpu {t3-t2:.3f}s dev {stats.device_ms:.1f}ms")
    g.close(); S.close()
    return ref, img

sc, cam_m, yfov = scenes.box_scene()
ref, img = check("box", sc, cam_m, yfov, 64, 48, 4, 4)
check("box", sc, cam_m, yfov, 64, 48, 16, 16)
sc, cam_m, yfov = scenes.box_scene(with_mirror=True)
check("box+mirror", sc, cam_m, yfov, 64, 48, 4, 16)
sc, cam_m, yfov = scenes.c1_scene(n_small=20)
check("c1", sc, cam_m, yfov, 160, 120, 1, 1)
check("c1", sc, cam_m, yfov, 80, 60, 4, 8)
sc, cam_m, yfov = scenes.interior_scene(detail=0.05)
check("interior", sc, cam_m, yfov, 96, 54, 4, 4)
