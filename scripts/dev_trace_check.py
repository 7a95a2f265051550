"""Developer check: GPU closest/any hit vs the CPU oracle on the C1 scene and random rays."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import oracle
from raytracer_amd import scenes, abi, api

def compare(name, sc, o, d):
    S = oracle.OracleScene(sc)
    t0 = time.time(); ref = S.trace_closest(o, d); t1 = time.time()
    G = api.Scene(sc)
    info = G.info()
    t2 = time.time(); got = G.trace_rays(o, d); t3 = time.time()
    hit_ref = ref["inst"] != abi.INVALID_ID
    hit_got = got["inst"] != abi.INVALID_ID
    same = (ref["inst"] == got["inst"]) & (ref["prim"] == got["prim"])
    bad = np.nonzero(~same)[0]
    both = hit_ref & hit_got & same
    rel = np.abs(ref["t"][both] - got["t"][both]) / np.maximum(np.abs(ref["t"][both]), 1e-30)
    print(f"{name}: n={len(o)} hits ref={hit_ref.sum()} gpu={hit_got.sum()} id-mismatch={len(bad)} max rel t={rel.max() if rel.size else 0:.3g} "
          f"exact t={np.mean(ref['t'][both]==got['t'][both]):.6f} oracle {t1-t0:.2f}s gpu {t3-t2:.3f}s nodes={info.n_nodes} depth={info.max_depth} leafmax={info.max_leaf_size} build={info.build_ms:.0f}ms")
    if len(bad):
        brute, tied = S.brute_closest(o[bad], d[bad])
        for k, i in enumerate(bad[:10]):
            print("   ray", i, "ref", ref[i], "gpu", got[i], "brute", brute[k], "tied", tied[k])
    # any-hit
    tmax = np.where(hit_ref, ref["t"] * 0.999, 5.0).astype(np.float32)
    tmax2 = np.where(hit_ref, ref["t"] * 1.001, 5.0).astype(np.float32)
    for tm in (tmax, tmax2):
        a = S.trace_any(o, d, tm); b = G.test_visibility(o, d, tm)
        print("   any-hit mismatches:", int((a != b).sum()), "occluded", int(a.sum()))
    G.close(); S.close()

sc, cam_m, yfov = scenes.c1_scene()
cam = api.camera_from_transform(scenes.x_fov_degrees(yfov, 640 / 480), cam_m)
cam_o = oracle.camera_from_transform(scenes.x_fov_degrees(yfov, 640 / 480), cam_m)
print("camera equal:", bytes(cam) == bytes(cam_o))
st = abi.make_settings(640, 480, geometry_aa=1, material_aa=1, jitter_mode=abi.JITTER_CENTER)
o, d = oracle.generate_primary(cam_o, st)
compare("c1-instanced primary", sc, o, d)
sc2, _, _ = scenes.c1_scene(soup=True, with_spheres=False)
compare("c1-soup primary", sc2, o, d)
ro, rd = scenes.random_rays(200000, (-1.5, -1.5, -1.5), (2.5, 2.5, 1.5), seed=3)
compare("c1-instanced random", sc, ro, rd)
sc3, cm3, yf3 = scenes.interior_scene(detail=0.08)
print("interior tris", sc3.n_triangles())
ro, rd = scenes.random_rays(200000, (0.5, 0.5, 0.5), (23, 8, 9), seed=4)
compare("interior random", sc3, ro, rd)
compare("two spheres", scenes.two_spheres_scene(), np.zeros((2, 3), np.float32), np.array([[-1, 0, 0], [1, 0, 0]], np.float32))
