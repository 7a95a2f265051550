"""Developer probe: GPU-vs-oracle id mismatches of bounce rays in the reduced interior, with the brute-force answer."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import oracle
from raytracer_amd import abi, api, scenes
from test_gpu_configs import bounce_rays, primary

sc, cam_m, yfov = scenes.interior_scene(detail=0.3)
S = oracle.OracleScene(sc)
o, d = primary(cam_m, yfov, 320, 180)
np.set_printoptions(precision=9, suppress=False, linewidth=200)
with api.Scene(sc) as g:
    got = g.trace_rays(o, d)
    for gen in range(3):
        o, d = bounce_rays(o, d, got, seed=10 + gen)
        got = g.trace_rays(o, d)
        ref = S.trace_closest(o, d)
        bad = np.nonzero((ref["prim"] != got["prim"]) | (ref["inst"] != got["inst"]))[0]
        brute, tied = S.brute_closest(o[bad], d[bad])
        for k, i in enumerate(bad):
            if tied[k] > 1:
                continue
            print("gen", gen, "ray", i, "o", o[i], "d", d[i])
            print("   oracle", ref[i], " gpu", got[i], " brute", brute[k], "tied", tied[k])
            m = sc.meshes[0]
            for name, h in (("oracle", ref[i]), ("gpu", got[i])):
                if h["prim"] != abi.INVALID_ID:
                    print("   ", name, "tri", m.vertices[m.indices[h["prim"]]].tolist())
            # the same ray alone, and through the per-ray any-hit query
            one = g.trace_rays(o[i:i + 1], d[i:i + 1])
            print("    alone:", one[0])
