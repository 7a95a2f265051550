"""GPU-vs-oracle coverage of the larger configurations of BASELINE.json at sizes the oracle finishes in seconds:
incoherent GI rays (bounce rays from the primary hits) on a reduced configs[4] cluster -- flattened and as instances --
and on a reduced configs[2] interior; configs[3]'s 3840x2160 frame assembled from the square sets of 2 / 4 / 8 ranks;
and a scene with a degenerate triangle, where the cooperative kernel and the reference's first-hit rule differ."""
import numpy as np
import pytest

import oracle
from raytracer_amd import abi, api, scenes
from test_gpu_parity import assert_hits_match

pytestmark = pytest.mark.gpu


def bounce_rays(o, d, hits, seed):
    """From every hit point a random direction on the side the ray came from (an incoherent second generation)."""
    hit = hits["inst"] != abi.INVALID_ID
    hp = (o[hit] + d[hit] * hits["t"][hit][:, None]).astype(np.float32)
    rng = np.random.RandomState(seed)
    nd = rng.normal(size=hp.shape)
    nd /= np.linalg.norm(nd, axis=1, keepdims=True)
    nd[(nd * d[hit]).sum(1) > 0] *= -1
    nd = nd.astype(np.float32)
    return (hp + nd * np.float32(1e-3)).astype(np.float32), nd


def primary(cam_m, yfov, W, H):
    cam = api.camera_from_transform(scenes.x_fov_degrees(yfov, W / H), cam_m)
    st = abi.make_settings(W, H, geometry_aa=1, material_aa=1, jitter_mode=abi.JITTER_CENTER)
    return oracle.generate_primary(cam, st)


@pytest.mark.parametrize("flatten", [True, False])
def test_gi_rays_on_a_reduced_cluster(flatten):
    """configs[4] at 38 x 5120 triangles: two generations of bounce rays against the oracle, ids with the tie rule, distances
    bit-exact; as one flattened mesh (cooperative walk over one tree) and as 38 instances (cooperative walk over top-level
    and mesh trees, fp64 inverse transforms)."""
    sc, cam_m, yfov = scenes.cluster_scene(38, 4, flatten=flatten)
    assert sc.n_triangles() * (1 if flatten else 38) == 38 * 5120
    S = oracle.OracleScene(sc)
    o, d = primary(cam_m, yfov, 320, 180)
    with api.Scene(sc) as g:
        got = g.trace_rays(o, d)
        assert_hits_match(S, got, o, d)
        assert (got["inst"] != abi.INVALID_ID).mean() > 0.5
        for gen in range(2):
            o, d = bounce_rays(o, d, got, seed=gen)
            got = g.trace_rays(o, d)
            assert_hits_match(S, got, o, d)
            assert 0.2 < (got["inst"] != abi.INVALID_ID).mean() < 1.0
            tmax = np.where(np.isfinite(got["t"]), got["t"] * np.where(np.arange(len(o)) % 2 == 0, 0.999, 1.001), 5.0).astype(np.float32)
            assert np.array_equal(g.test_visibility(o, d, tmax), S.trace_any(o, d, tmax))


def test_gi_rays_on_a_reduced_interior():
    sc, cam_m, yfov = scenes.interior_scene(detail=0.3)
    S = oracle.OracleScene(sc)
    o, d = primary(cam_m, yfov, 320, 180)
    with api.Scene(sc) as g:
        got = g.trace_rays(o, d)
        assert_hits_match(S, got, o, d)
        assert (got["inst"] != abi.INVALID_ID).all()      # a closed room
        for gen in range(3):
            o, d = bounce_rays(o, d, got, seed=10 + gen)
            got = g.trace_rays(o, d)
            assert_hits_match(S, got, o, d)


def test_c4_frame_from_the_square_sets_of_2_4_8_ranks_is_bit_identical():
    """configs[3]: the interior at 3840x2160 (one sample per pixel here), the frame cut into 32x32 squares over G = 2, 4, 8
    ranks, each rank's squares rendered into its compact buffer and the buffers written back by the gather's scatter
    kernel: the same bits as the single pass."""
    import torch
    sc, cam_m, yfov = scenes.interior_scene(scenes.C3_DETAIL)
    W, H, T = 3840, 2160, 32
    cam = api.camera_from_transform(scenes.x_fov_degrees(yfov, W / H), cam_m)
    st = abi.make_settings(W, H, geometry_aa=1, material_aa=1, max_path_length=9, seed=1)
    dev = torch.device("cuda", 0)
    r = api.Renderer()
    with api.Scene(sc) as g:
        full = torch.zeros((H, W, 3), dtype=torch.float32, device=dev)
        stats = r.render_device(g, cam, st, full)
        assert stats.primary_rays == W * H and float(full[torch.isfinite(full)].mean()) > 0.01
        for world in (2, 4, 8):
            stride = api.tile_buffer_floats(st, T, 0, world)
            d_all = torch.zeros(world * stride, dtype=torch.float32, device=dev)
            rays = 0
            for rank in range(world):
                s = r.render_tiles_device(g, cam, st, T, rank, world, d_all[rank * stride:])
                rays += s.primary_rays
            assert rays == W * H
            frame = torch.full((H, W, 3), -1.0, dtype=torch.float32, device=dev)
            api.scatter_tiles_device(st, T, world, d_all, frame, torch.cuda.current_stream())
            torch.cuda.synchronize()
            assert torch.equal(frame.view(torch.int32), full.view(torch.int32))
            del d_all, frame


@pytest.mark.parametrize("flatten", [True, False])
@pytest.mark.parametrize("distance", [30.0, 300.0, 1200.0])
def test_rays_from_far_away_find_what_a_search_without_boxes_finds(flatten, distance):
    """Rays that start `distance` units outside the reduced cluster, aimed at points inside it.  The kernels' box tests only
    prune, so the answer has to be the closest triangle of a search over all triangles without any box (the oracle's brute
    force): beyond a few hundred units the rounding of a plain box test comes close to the 1e-4 the reference pads its boxes
    with, and the waves take the widened box decode.  (Beyond about 1 600 units the fp32 triangle test itself answers for
    rays that pass a triangle by more than that padding -- 2^-24 of the distance -- and no box test, the reference's
    included, is conservative against it: that is the range the library supports.)"""
    sc, cam_m, yfov = scenes.cluster_scene(38, 3, flatten=flatten)
    S = oracle.OracleScene(sc)
    rng = np.random.RandomState(7)
    n = 3000
    o0, d0 = primary(cam_m, yfov, 64, 48)
    with api.Scene(sc) as g:
        first = g.trace_rays(o0, d0)
        hit = first["inst"] != abi.INVALID_ID
        pts = (o0[hit] + d0[hit] * first["t"][hit][:, None]).astype(np.float64)      # points on the surfaces
        target = pts[rng.randint(0, len(pts), n)]
        dirs = rng.normal(size=(n, 3))
        dirs /= np.linalg.norm(dirs, axis=1, keepdims=True)
        o = (target - dirs * distance).astype(np.float32)
        d = (target - o.astype(np.float64))
        d = (d / np.linalg.norm(d, axis=1, keepdims=True)).astype(np.float32)
        got = g.trace_rays(o, d)
        brute, tied = S.brute_closest(o, d)
        assert np.array_equal(brute["inst"] != abi.INVALID_ID, got["inst"] != abi.INVALID_ID)
        assert np.array_equal(brute["prim"], got["prim"]) and np.array_equal(brute["inst"], got["inst"])
        assert np.array_equal(brute["t"], got["t"])
        assert (got["inst"] != abi.INVALID_ID).mean() > 0.5
        tmax = np.where(np.isfinite(got["t"]), got["t"] * np.float32(1.0001), 1.0).astype(np.float32)
        occ = g.test_visibility(o, d, tmax)
        assert np.array_equal(occ != 0, got["inst"] != abi.INVALID_ID)


def test_degenerate_triangle_nan_and_infinite_distances():
    """A triangle so small that its determinant is a denormal number makes Triangle::intersect (math/Triangle.h:34-101)
    divide by it: 1 / pDet overflows to infinity and t = p0Det * inf is +inf, or NaN when the ray starts in the triangle's
    plane at its vertex (0 * inf).  The reference keeps such a hit by its first-hit rule (`!bestHit || bestHit->t > t`):
    a NaN hit met first is never replaced, met later it never replaces -- its answer depends on its tree.  The cooperative
    kernel defines the result tree-independently: a hit needs t <= the best so far, which NaN never satisfies, so the NaN
    candidate is dropped and the ray goes on to what lies behind; t = +inf is an ordinary (farthest possible) hit on both
    sides.  A ray that starts exactly in a triangle's plane gets t = -0 from the reference's arithmetic, which is not < 0 and
    therefore a hit at distance zero; the kernel reports it as +0."""
    tiny = np.float32(1e-20)
    v = np.array([[-4, -4, -1], [4, -4, -1], [4, 4, -1], [-4, 4, -1],          # floor at z = -1
                  [0, 0, 0], [tiny, 0, 0], [0, tiny, 0]], np.float32)          # the degenerate triangle in the plane z = 0
    f = np.array([[0, 1, 2], [0, 2, 3], [4, 5, 6]], np.uint32)
    n = np.tile(np.array([[0, 0, 1]], np.float32), (len(v), 1))
    sc = abi.SceneData()
    sc.meshes = [abi.MeshData(v, n, f)]
    sc.materials = [abi.MaterialData(abi.MAT_DIFFUSE, (0.7, 0.7, 0.7))]
    sc.instances = [abi.InstanceData(0, 0, np.eye(4), np.eye(4))]
    o = np.array([[0, 0, 0], [0, 0, 1], [0, 0, 1], [0, 0, -0.5], [1, 1, 1], [1, 1, -1], [1, 1, -1]], np.float32)
    d = np.array([[0, 0, -1], [0, 0, -1], [0, 0, 1], [0, 0, 1], [0, 0, -1], [0.6, 0, -0.8], [0.6, 0, 0.8]], np.float32)
    S = oracle.OracleScene(sc)
    ref = S.trace_closest(o, d)
    with api.Scene(sc) as g:
        got = g.trace_rays(o, d)
    # ray 0 starts at the tiny triangle's vertex: NaN candidate, then the floor at t = 1
    assert got["prim"][0] in (0, 1) and got["t"][0] == 1.0
    assert (np.isnan(ref["t"][0]) and ref["prim"][0] == 2) or (ref["prim"][0] == got["prim"][0] and ref["t"][0] == 1.0)
    # ray 1 passes through it from above: t = +inf there, the floor at t = 2 is closer
    assert got["prim"][1] == ref["prim"][1] and got["t"][1] == ref["t"][1] == 2.0
    # ray 2 leaves upwards: the tiny triangle lies behind it (t = -inf): miss
    assert got["inst"][2] == ref["inst"][2] == abi.INVALID_ID
    # ray 3 comes from below with nothing else ahead: the infinitely distant hit is the answer on both sides
    assert got["prim"][3] == ref["prim"][3] == 2 and np.isinf(got["t"][3]) and np.isinf(ref["t"][3])
    # ray 4 is an ordinary ray
    assert got["prim"][4] == ref["prim"][4] and got["t"][4] == ref["t"][4] == 2.0
    # rays 5 and 6 start exactly in the floor's plane: t = +0 or -0 (neither is < 0, so Triangle::intersect accepts both);
    # a hit at distance zero on both sides, whatever the sign of the zero
    for k in (5, 6):
        assert ref["inst"][k] == 0 and ref["t"][k] == 0.0
        assert got["prim"][k] in (0, 1) and got["t"][k] == 0.0 and got["inst"][k] == 0
    assert np.signbit(ref["t"][5]) or np.signbit(ref["t"][6])     # one of the two is the -0 case
