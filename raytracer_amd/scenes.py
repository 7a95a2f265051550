"""Procedural, seeded scenes for the benchmark configurations and the parity tests.

The reference ships no mesh assets (scenegen.py:391 loads an OBJ from the author's disk), so
every workload is generated here.  Scene *recipes* follow the reference where it has one:
`c1_scene` restates generateSceneOneHugeManySmallLongInterval + generateScene
(scenegen.py:340-452): one mesh instanced 101 times with TRS nodes, a camera at
(0.5, 0.5, -3) rotated 180 degrees about Y with yfov 0.40, and a point light of intensity 100.
"""
from __future__ import annotations

import math

import numpy as np

from .abi import (MAT_ADD, MAT_CONSTMIX, MAT_DIFFUSE, MAT_EMISSIVE, MAT_FRESNELMIX, MAT_GLASS, MAT_GLOSSY, MAT_TRANSPARENT, SHAPE_SPHERE,
                  InstanceData, MaterialData, MeshData, SceneData)


# ---------------------------------------------------------------------------------------
# transforms (row-major 4x4 float64, like Transformation's Matrix4d)
# ---------------------------------------------------------------------------------------
def translate(x, y, z):
    m = np.eye(4)
    m[:3, 3] = [x, y, z]
    return m


def scale(x, y, z):
    return np.diag([x, y, z, 1.0]).astype(np.float64)


def rotate_quaternion(i, j, k, r):
    """Transformation::rotateQuaternion (math/Transformation.cpp:104-117)."""
    s = 1.0 / (r * r + i * i + j * j + k * k)
    return np.array([
        [1 - 2 * s * (j * j + k * k), 2 * s * (i * j - k * r), 2 * s * (i * k + j * r), 0],
        [2 * s * (i * j + k * r), 1 - 2 * s * (i * i + k * k), 2 * s * (j * k - i * r), 0],
        [2 * s * (i * k - j * r), 2 * s * (j * k + i * r), 1 - 2 * s * (i * i + j * j), 0],
        [0, 0, 0, 1]], dtype=np.float64)


def trs(translation, quat_xyzw, scl):
    """glTF node matrix T * R * S and its inverse."""
    t = translate(*translation)
    r = rotate_quaternion(*quat_xyzw)
    s = scale(*scl)
    m = t @ r @ s
    inv = scale(1.0 / scl[0], 1.0 / scl[1], 1.0 / scl[2]) @ r.T @ translate(-translation[0], -translation[1], -translation[2])
    return m, inv


def look_at(eye, target, up=(0.0, 1.0, 0.0)):
    """Object->world matrix of a camera at `eye` looking at `target` (camera looks along -z)."""
    eye = np.asarray(eye, dtype=np.float64)
    f = np.asarray(target, dtype=np.float64) - eye
    f /= np.linalg.norm(f)
    upv = np.asarray(up, dtype=np.float64)
    r = np.cross(f, upv)
    r /= np.linalg.norm(r)
    u = np.cross(r, f)
    m = np.eye(4)
    m[:3, 0], m[:3, 1], m[:3, 2], m[:3, 3] = r, u, -f, eye
    return m


def x_fov_degrees(yfov_rad, aspect):
    """io/GLTF.cpp:598-599."""
    return math.degrees(math.atan(math.tan(yfov_rad / 2.0) * aspect) * 2.0)


# ---------------------------------------------------------------------------------------
# meshes
# ---------------------------------------------------------------------------------------
def vertex_normals(v, f):
    v = v.astype(np.float64)
    fn = np.cross(v[f[:, 1]] - v[f[:, 0]], v[f[:, 2]] - v[f[:, 0]])
    n = np.zeros_like(v)
    for k in range(3):
        np.add.at(n, f[:, k], fn)
    ln = np.linalg.norm(n, axis=1, keepdims=True)
    ln[ln == 0] = 1.0
    return (n / ln).astype(np.float32)


def uv_sphere(segments=32, rings=16, radius=1.0):
    """Closed UV sphere: segments*2*(rings-1) triangles (32x16 -> 960)."""
    verts = [(0.0, radius, 0.0)]
    for r in range(1, rings):
        phi = math.pi * r / rings
        for s in range(segments):
            th = 2.0 * math.pi * s / segments
            verts.append((radius * math.sin(phi) * math.cos(th), radius * math.cos(phi), radius * math.sin(phi) * math.sin(th)))
    verts.append((0.0, -radius, 0.0))
    faces = []
    for s in range(segments):
        faces.append((0, 1 + (s + 1) % segments, 1 + s))
    for r in range(rings - 2):
        a0 = 1 + r * segments
        b0 = a0 + segments
        for s in range(segments):
            s1 = (s + 1) % segments
            faces.append((a0 + s, a0 + s1, b0 + s))
            faces.append((a0 + s1, b0 + s1, b0 + s))
    last = len(verts) - 1
    a0 = 1 + (rings - 2) * segments
    for s in range(segments):
        faces.append((last, a0 + s, a0 + (s + 1) % segments))
    v = np.asarray(verts, dtype=np.float32)
    f = np.asarray(faces, dtype=np.uint32)
    n = (v / np.linalg.norm(v, axis=1, keepdims=True)).astype(np.float32)
    return MeshData(v, n, f)


def icosphere(subdiv):
    t = (1.0 + math.sqrt(5.0)) / 2.0
    v = np.array([[-1, t, 0], [1, t, 0], [-1, -t, 0], [1, -t, 0], [0, -1, t], [0, 1, t], [0, -1, -t], [0, 1, -t],
                  [t, 0, -1], [t, 0, 1], [-t, 0, -1], [-t, 0, 1]], dtype=np.float64)
    v /= np.linalg.norm(v, axis=1, keepdims=True)
    f = np.array([[0, 11, 5], [0, 5, 1], [0, 1, 7], [0, 7, 10], [0, 10, 11], [1, 5, 9], [5, 11, 4], [11, 10, 2],
                  [10, 7, 6], [7, 1, 8], [3, 9, 4], [3, 4, 2], [3, 2, 6], [3, 6, 8], [3, 8, 9], [4, 9, 5],
                  [2, 4, 11], [6, 2, 10], [8, 6, 7], [9, 8, 1]], dtype=np.int64)
    for _ in range(subdiv):
        nv = v.shape[0]
        e = np.concatenate([f[:, [0, 1]], f[:, [1, 2]], f[:, [2, 0]]], axis=0)
        e.sort(axis=1)
        key = e[:, 0] * nv + e[:, 1]
        uniq, inv = np.unique(key, return_inverse=True)
        mid = v[uniq // nv] + v[uniq % nv]
        mid /= np.linalg.norm(mid, axis=1, keepdims=True)
        v = np.concatenate([v, mid], axis=0)
        nf = f.shape[0]
        m01, m12, m20 = nv + inv[:nf], nv + inv[nf:2 * nf], nv + inv[2 * nf:]
        f = np.concatenate([
            np.stack([f[:, 0], m01, m20], axis=1), np.stack([f[:, 1], m12, m01], axis=1),
            np.stack([f[:, 2], m20, m12], axis=1), np.stack([m01, m12, m20], axis=1)], axis=0)
    return v, f


def displaced_icosphere(subdiv, seed=7, octaves=6, amplitude=0.22):
    """Closed, high-curvature 'bunny/dragon-class' blob: 20*4^subdiv triangles (8 -> 1 310 720)."""
    v, f = icosphere(subdiv)
    rng = np.random.RandomState(seed)
    r = np.ones(v.shape[0])
    for k in range(octaves):
        for _ in range(3):
            d = rng.normal(size=3)
            d /= np.linalg.norm(d)
            freq = 1.9 * (2.0 ** k)
            r += (amplitude / (2.0 ** k) / 3.0) * np.sin(freq * (v @ d) + rng.uniform(0, 2 * math.pi))
    v = v * r[:, None]
    vf = v.astype(np.float32)
    ff = f.astype(np.uint32)
    return MeshData(vf, vertex_normals(vf, ff), ff)


def grid_quad(p0, du, dv, nu, nv):
    """Tessellated parallelogram p0 + s*du + t*dv, normal du x dv."""
    p0, du, dv = (np.asarray(a, dtype=np.float64) for a in (p0, du, dv))
    s = np.linspace(0, 1, nu + 1)
    t = np.linspace(0, 1, nv + 1)
    ss, tt = np.meshgrid(s, t, indexing="xy")
    v = p0[None, :] + ss.reshape(-1, 1) * du[None, :] + tt.reshape(-1, 1) * dv[None, :]
    idx = (np.arange(nv)[:, None] * (nu + 1) + np.arange(nu)[None, :]).reshape(-1)
    f = np.concatenate([np.stack([idx, idx + 1, idx + nu + 1], 1), np.stack([idx + 1, idx + nu + 2, idx + nu + 1], 1)], 0)
    return v, f


def cylinder(base, radius, height, segments, stacks, flute=0.0):
    base = np.asarray(base, dtype=np.float64)
    th = np.linspace(0, 2 * math.pi, segments, endpoint=False)
    ys = np.linspace(0, height, stacks + 1)
    rr = radius * (1.0 + flute * np.cos(8 * th))
    v = np.stack([
        (rr[None, :] * np.cos(th)[None, :]).repeat(stacks + 1, 0),
        ys[:, None].repeat(segments, 1),
        (rr[None, :] * np.sin(th)[None, :]).repeat(stacks + 1, 0)], axis=-1).reshape(-1, 3) + base[None, :]
    a = (np.arange(stacks)[:, None] * segments + np.arange(segments)[None, :]).reshape(-1)
    a1 = (np.arange(stacks)[:, None] * segments + (np.arange(segments)[None, :] + 1) % segments).reshape(-1)
    f = np.concatenate([np.stack([a, a + segments, a1], 1), np.stack([a1, a + segments, a1 + segments], 1)], 0)
    return v, f


def _merge(parts):
    """parts: list of (v, f, material id) -> one MeshData with tri_material."""
    vs, fs, ms = [], [], []
    off = 0
    for v, f, mat in parts:
        vs.append(np.asarray(v, dtype=np.float64))
        fs.append(np.asarray(f, dtype=np.int64) + off)
        ms.append(np.full(len(f), mat, dtype=np.uint32))
        off += len(v)
    v = np.concatenate(vs).astype(np.float32)
    f = np.concatenate(fs).astype(np.uint32)
    return MeshData(v, vertex_normals(v, f), f, np.concatenate(ms))


# ---------------------------------------------------------------------------------------
# scenes
# ---------------------------------------------------------------------------------------
def two_spheres_scene():
    """raytracer/test/BVH.cpp:18-49: unit spheres at x = -2 and x = +2."""
    sc = SceneData()
    sc.materials = [MaterialData(MAT_DIFFUSE, (0.7, 0.7, 0.7))]
    sc.instances = [InstanceData(SHAPE_SPHERE, 0, translate(-2, 0, 0), translate(2, 0, 0)),
                    InstanceData(SHAPE_SPHERE, 0, translate(2, 0, 0), translate(-2, 0, 0))]
    return sc


def c1_scene(seed=0, n_small=100, with_spheres=True, soup=False):
    """Config 1: 1 big + n_small small instances of a 960-triangle UV sphere (scenegen.py:340-387),
    plus 16 analytic unit spheres, one point light; returns (scene, camera transform, yfov)."""
    rng = np.random.RandomState(seed)
    mesh = uv_sphere(32, 16)
    sc = SceneData()
    sc.materials = [MaterialData(MAT_DIFFUSE, (0.7, 0.7, 0.7))]
    q = (0.0, 1.0, 0.0, 0.0)
    nodes = [((0.0, 0.0, 0.0), q, (0.85, 0.85, 0.85))]
    for i in range(n_small):
        nodes.append(((float(i), rng.random_sample() * 0.4, rng.random_sample() * 0.4), q, (0.025, 0.025, 0.025)))
    if soup:
        parts = []
        for tr, qq, s in nodes:
            m, _ = trs(tr, qq, s)
            v = mesh.vertices.astype(np.float64) @ m[:3, :3].T + m[:3, 3]
            parts.append((v, mesh.indices, 0))
        sc.meshes = [_merge(parts)]
        sc.instances = [InstanceData(0, 0, np.eye(4), np.eye(4))]
    else:
        sc.meshes = [mesh]
        for tr, qq, s in nodes:
            m, inv = trs(tr, qq, s)
            sc.instances.append(InstanceData(0, 0, m, inv))
    if with_spheres:
        for k in range(16):
            ang = 2.0 * math.pi * k / 16.0
            pos = (0.5 + 0.3 * math.cos(ang), 0.5 + 0.3 * math.sin(ang), -1.2 + 0.2 * math.sin(3 * ang))
            m, inv = trs(pos, (0.0, 0.0, 0.0, 1.0), (0.05, 0.05, 0.05))
            sc.instances.append(InstanceData(SHAPE_SPHERE, 0, m, inv))
    sc.point_lights = [((0.5, 0.5, -3.0), (1.0, 1.0, 1.0), 100.0)]
    cam_m, _ = trs((0.5, 0.5, -3.0), q, (1.0, 1.0, 1.0))
    return sc, cam_m, 0.40


def blob_scene(subdiv=8, seed=7):
    """Config 2: fractal-displaced icosphere filling most of the frame height (~45 % of the pixels hit)."""
    mesh = displaced_icosphere(subdiv, seed)
    sc = SceneData()
    sc.materials = [MaterialData(MAT_DIFFUSE, (0.7, 0.7, 0.7))]
    mesh.tri_material = None
    sc.meshes = [mesh]
    sc.instances = [InstanceData(0, 0, np.eye(4), np.eye(4))]
    sc.point_lights = [((3.0, 4.0, 5.0), (1.0, 1.0, 1.0), 600.0)]
    yfov = 0.6
    radius = float(np.linalg.norm(mesh.vertices, axis=1).mean())
    dist = radius / (0.95 * math.tan(yfov / 2.0))   # mean radius spans 95 % of the frame height
    cam_m = look_at((0.35 * dist, 0.25 * dist, dist * 0.9), (0.0, 0.0, 0.0))
    return sc, cam_m, yfov


def blob_room_scene(subdiv=8, seed=7, wall_grid=24):
    """The ~1 M-triangle scene of the path-traced throughput figure: the config-2 blob (20 * 4^subdiv triangles, about
    70 % of its surface diffuse grey and 30 % glossy in bands, like the material mix of config 3) standing in a closed
    room (six tessellated walls, a glossy floor) lit by an area light under the ceiling and a point light, seen from
    the camera of config 2.  One mesh, per-triangle materials.  Paths bounce between the blob and the walls, so
    extension and shadow rays are incoherent and most of them have to search the blob's tree."""
    blob = displaced_icosphere(subdiv, seed)
    v, f = blob.vertices.astype(np.float64), blob.indices
    cy = v[f].mean(axis=1)[:, 1]
    glossy = np.sin(7.0 * cy + 0.4) > 0.45
    ymin = float(v[:, 1].min())
    R = 6.0
    y0, y1 = ymin - 0.05, ymin - 0.05 + 2 * R
    g = wall_grid
    parts = [(v, f[~glossy], 0), (v, f[glossy], 1),
             (*grid_quad((-R, y0, -R), (0, 0, 2 * R), (2 * R, 0, 0), g, g), 1),        # floor (glossy), normal +y
             (*grid_quad((-R, y1, -R), (2 * R, 0, 0), (0, 0, 2 * R), g, g), 0),        # ceiling, normal -y
             (*grid_quad((-R, y0, -R), (2 * R, 0, 0), (0, 2 * R, 0), g, g), 2),        # z = -R wall, normal +z
             (*grid_quad((-R, y0, R), (0, 2 * R, 0), (2 * R, 0, 0), g, g), 3),         # z = +R wall, normal -z
             (*grid_quad((-R, y0, -R), (0, 2 * R, 0), (0, 0, 2 * R), g, g), 0),        # x = -R wall, normal +x
             (*grid_quad((R, y0, -R), (0, 0, 2 * R), (0, 2 * R, 0), g, g), 0)]         # x = +R wall, normal -x
    mesh = _merge(parts)
    # the blob's two parts share one vertex array: its smooth normals must come from all of its faces
    nb = len(v)
    mesh.normals[:nb] = blob.normals
    mesh.normals[nb:2 * nb] = blob.normals
    sc = SceneData()
    sc.meshes = [mesh]
    sc.materials = [MaterialData(MAT_DIFFUSE, (0.7, 0.7, 0.7)), MaterialData(MAT_GLOSSY, (0.9, 0.9, 0.9), roughness=0.2),
                    MaterialData(MAT_DIFFUSE, (0.7, 0.25, 0.2)), MaterialData(MAT_DIFFUSE, (0.2, 0.3, 0.7))]
    sc.instances = [InstanceData(0, 0, np.eye(4), np.eye(4))]
    sc.area_lights = [((-1.5, y1 - 0.05, -1.0), (1.5, y1 - 0.05, -1.0), (0.0, y1 - 0.05, 1.6), (1.0, 0.95, 0.85), 700.0)]
    sc.point_lights = [((3.0, 4.0, 5.0), (0.9, 0.95, 1.0), 400.0)]
    yfov = 0.6
    radius = float(np.linalg.norm(blob.vertices, axis=1).mean())
    dist = radius / (0.95 * math.tan(yfov / 2.0))
    cam_m = look_at((0.35 * dist, 0.25 * dist, dist * 0.9), (0.0, 0.0, 0.0))
    return sc, cam_m, yfov


def cluster_scene(copies=38, subdiv=8, flatten=True, seed=7):
    """Config 5: `copies` translated copies of the config-2 mesh on a jittered lattice (38 copies of
    1 310 720 triangles = 49.8 M), either flattened into one triangle soup (what --soupify makes of
    it; triangle index = copy * n_triangles + index in the mesh) or kept as instances of one mesh.
    Translations are multiples of 1/4, so the flattened vertices are exact in float for this size.
    Returns (scene, camera matrix, yfov); the camera sits inside the cluster."""
    mesh = displaced_icosphere(subdiv, seed)
    rng = np.random.RandomState(seed + 1)
    side = int(math.ceil(copies ** (1.0 / 3.0)))
    cells = [(i, j, k) for i in range(side) for j in range(side) for k in range(side)]
    rng.shuffle(cells)
    cells = cells[:copies]
    pitch = 3.0
    offs = np.array([[c[0] * pitch + rng.randint(-2, 3) * 0.25, c[1] * pitch + rng.randint(-2, 3) * 0.25,
                      c[2] * pitch + rng.randint(-2, 3) * 0.25] for c in cells], dtype=np.float64)
    sc = SceneData()
    sc.materials = [MaterialData(MAT_DIFFUSE, (0.7, 0.7, 0.7))]
    sc.point_lights = [((-3.0, 12.0, -3.0), (1.0, 1.0, 1.0), 3000.0)]
    if flatten:
        nv, nt = mesh.vertices.shape[0], mesh.indices.shape[0]
        v = np.empty((copies * nv, 3), dtype=np.float32)
        n = np.empty((copies * nv, 3), dtype=np.float32)
        idx = np.empty((copies * nt, 3), dtype=np.uint32)
        for c in range(copies):
            v[c * nv:(c + 1) * nv] = (mesh.vertices.astype(np.float64) + offs[c][None, :]).astype(np.float32)
            n[c * nv:(c + 1) * nv] = mesh.normals
            idx[c * nt:(c + 1) * nt] = mesh.indices + np.uint32(c * nv)
        sc.meshes = [MeshData(v, n, idx)]
        sc.instances = [InstanceData(0, 0, np.eye(4), np.eye(4))]
    else:
        mesh.tri_material = None
        sc.meshes = [mesh]
        sc.instances = [InstanceData(0, 0, translate(*offs[c]), translate(*(-offs[c]))) for c in range(copies)]
    centre = (side - 1) * pitch / 2.0
    cam_m = look_at((centre + 0.4, centre + 1.5, centre + 0.2), (centre + pitch, centre - 0.5, centre + pitch * 0.8))
    return sc, cam_m, 0.9


C3_DETAIL = 1.78


def interior_scene(detail=1.0, seed=1):
    """Configs 3/4: closed colonnaded atrium ('Sponza-class').  The triangle count grows linearly
    with detail: 1.0 gives 140 672, C3_DETAIL (1.78) gives ~250 k triangles,
    ~70 % of them diffuse (albedo 0.7) and ~30 % glossy (roughness 0.2); one area light under the
    ceiling and one point light.  Materials: 0 diffuse grey, 1 glossy, 2 diffuse red, 3 diffuse
    blue."""
    rng = np.random.RandomState(seed)
    L, Wd, H = 24.0, 10.0, 9.0  # x in [0,L], z in [0,Wd], y in [0,H]
    g = max(2, int(round(36 * math.sqrt(detail))))
    parts = []
    parts.append((*grid_quad((0, 0, 0), (0, 0, Wd), (L, 0, 0), g, 2 * g), 1))        # floor (glossy), normal +y
    parts.append((*grid_quad((0, H, 0), (L, 0, 0), (0, 0, Wd), 2 * g, g), 0))        # ceiling, normal -y
    parts.append((*grid_quad((0, 0, 0), (L, 0, 0), (0, H, 0), 2 * g, g), 2))         # z=0 wall, normal +z
    parts.append((*grid_quad((0, 0, Wd), (0, H, 0), (L, 0, 0), g, 2 * g), 3))        # z=Wd wall, normal -z
    parts.append((*grid_quad((0, 0, 0), (0, H, 0), (0, 0, Wd), g, g), 0))            # x=0 wall, normal +x
    parts.append((*grid_quad((L, 0, 0), (0, 0, Wd), (0, H, 0), g, g), 0))            # x=L wall, normal -x
    ncol = 8
    seg = max(8, int(round(40 * math.sqrt(detail))))
    stk = max(4, int(round(56 * math.sqrt(detail))))
    for row, z in enumerate((2.2, Wd - 2.2)):
        for c in range(ncol):
            x = 2.0 + c * (L - 4.0) / (ncol - 1)
            mat = 1 if (c + row) % 3 == 0 else 0
            parts.append((*cylinder((x, 0.0, z), 0.42, H * 0.72, seg, stk, flute=0.06), mat))
            v, f = grid_quad((x - 0.7, H * 0.72 + 0.3, z - 0.7), (0, 0, 1.4), (1.4, 0, 0), 4, 4)  # capital slab (top face)
            parts.append((v, f, 0))
            v, f = grid_quad((x - 0.7, H * 0.72 + 0.01, z - 0.7), (1.4, 0, 0), (0, 0, 1.4), 4, 4)  # capital slab (bottom face)
            parts.append((v, f, 0))
    us = uv_sphere(max(8, int(48 * math.sqrt(detail))), max(4, int(24 * math.sqrt(detail))))
    for k in range(6):
        c = np.array([3.5 + 3.4 * k, 0.9 + 0.25 * (k % 2), Wd / 2 + 1.2 * math.sin(1.3 * k)])
        rad = 0.9 + 0.25 * (k % 2)
        parts.append((us.vertices.astype(np.float64) * rad + c[None, :], us.indices, 1 if k % 2 == 0 else (2 if k % 3 == 0 else 3)))
    # wavy drapes along the z=Wd wall
    nd = max(4, int(round(60 * math.sqrt(detail))))
    for k in range(4):
        x0 = 3.0 + 5.0 * k
        v, f = grid_quad((x0, 1.0, Wd - 0.35), (0, 5.5, 0), (3.0, 0, 0), nd, nd)
        v = v.copy()
        v[:, 2] += 0.18 * np.sin(9.0 * v[:, 0] + rng.uniform(0, 6.28)) * (0.3 + (v[:, 1] - 1.0) / 5.5)
        parts.append((v, f, 2 if k % 2 == 0 else 3))
    mesh = _merge(parts)
    sc = SceneData()
    sc.meshes = [mesh]
    sc.materials = [MaterialData(MAT_DIFFUSE, (0.7, 0.7, 0.7)), MaterialData(MAT_GLOSSY, (0.9, 0.9, 0.9), roughness=0.2),
                    MaterialData(MAT_DIFFUSE, (0.7, 0.25, 0.2)), MaterialData(MAT_DIFFUSE, (0.2, 0.3, 0.7))]
    sc.instances = [InstanceData(0, 0, np.eye(4), np.eye(4))]
    sc.area_lights = [((L / 2 - 2.5, H - 0.05, Wd / 2 - 1.5), (L / 2 + 2.5, H - 0.05, Wd / 2 - 1.5), (L / 2, H - 0.05, Wd / 2 + 2.0),
                       (1.0, 0.95, 0.85), 900.0)]
    sc.point_lights = [((4.0, 6.5, Wd / 2), (0.9, 0.95, 1.0), 500.0)]
    cam_m = look_at((1.6, 3.2, Wd / 2 + 0.4), (L - 4.0, 2.6, Wd / 2 - 0.3))
    return sc, cam_m, 0.9


def box_scene(with_mirror=False):
    """Small closed diffuse+glossy box with one point light and one area light (integrator tests)."""
    parts = []
    n = 6
    parts.append((*grid_quad((0, 0, 0), (0, 0, 4), (4, 0, 0), n, n), 0))      # floor
    parts.append((*grid_quad((0, 4, 0), (4, 0, 0), (0, 0, 4), n, n), 0))      # ceiling
    parts.append((*grid_quad((0, 0, 0), (4, 0, 0), (0, 4, 0), n, n), 2))      # back z=0
    parts.append((*grid_quad((0, 0, 4), (0, 4, 0), (4, 0, 0), n, n), 0))      # front z=4
    parts.append((*grid_quad((0, 0, 0), (0, 4, 0), (0, 0, 4), n, n), 3))      # left
    parts.append((*grid_quad((4, 0, 0), (0, 0, 4), (0, 4, 0), n, n), 1))      # right (glossy)
    us = uv_sphere(16, 8)
    parts.append((us.vertices.astype(np.float64) * 0.7 + np.array([1.4, 0.7, 1.6]), us.indices, 1))
    parts.append((us.vertices.astype(np.float64) * 0.5 + np.array([2.8, 0.5, 2.2]), us.indices, 4 if with_mirror else 0))
    mesh = _merge(parts)
    sc = SceneData()
    sc.meshes = [mesh]
    sc.materials = [MaterialData(MAT_DIFFUSE, (0.7, 0.7, 0.7)), MaterialData(MAT_GLOSSY, (0.9, 0.9, 0.9), roughness=0.25),
                    MaterialData(MAT_DIFFUSE, (0.7, 0.2, 0.2)), MaterialData(MAT_DIFFUSE, (0.2, 0.7, 0.2)),
                    MaterialData(MAT_GLOSSY, (0.95, 0.95, 0.95), roughness=0.0),
                    MaterialData(MAT_EMISSIVE, (1.0, 0.9, 0.8), intensity=2.0)]
    sc.instances = [InstanceData(0, 0, np.eye(4), np.eye(4))]
    sc.area_lights = [((1.5, 3.95, 1.5), (2.5, 3.95, 1.5), (2.0, 3.95, 2.5), (1.0, 1.0, 1.0), 60.0)]
    sc.point_lights = [((0.8, 3.0, 3.2), (1.0, 1.0, 1.0), 40.0)]
    cam_m = look_at((2.0, 2.0, 3.9), (2.0, 1.6, 0.0))
    return sc, cam_m, 0.9


def coat_scene():
    """The box of box_scene with the 'next' closures: a clear-coated sphere (Fresnel mix of a mirror
    over red diffuse), a coated floor (Fresnel mix nested in a constant mix), an all-glossy Fresnel
    mix (no variance where the mirror child is certain) and a transparent pane in front of the camera."""
    sc, cam_m, yfov = box_scene()
    base = len(sc.materials)
    sc.materials += [
        MaterialData(MAT_FRESNELMIX, roughness=1.45, first=2, second=4),          # base+0: coat: diffuse red under a mirror
        MaterialData(MAT_FRESNELMIX, roughness=1.33, first=0, second=1),          # base+1: grey diffuse under rough glossy
        MaterialData(MAT_CONSTMIX, mix_factor=0.35, first=base + 1, second=3),     # base+2: nested
        MaterialData(MAT_TRANSPARENT),                                             # base+3
        MaterialData(MAT_FRESNELMIX, roughness=1.6, first=4, second=4),            # base+4: mirror either way (no variance)
    ]
    mesh = sc.meshes[0]
    tm = mesh.tri_material.copy()
    n = 6
    quad = 2 * n * n
    tm[0:quad] = base + 2                     # floor
    tm[5 * quad:6 * quad] = base + 4          # right wall
    nsph = (len(tm) - 6 * quad) // 2
    tm[6 * quad:6 * quad + nsph] = base + 0   # first sphere
    pane_v, pane_f = grid_quad((1.0, 0.8, 3.2), (2.0, 0.0, 0.0), (0.0, 1.8, 0.0), 2, 2)
    pane_v = np.asarray(pane_v, dtype=np.float32)
    pane_f = np.asarray(pane_f, dtype=np.uint32)
    nv = len(mesh.vertices)
    sc.meshes = [MeshData(np.concatenate([mesh.vertices, pane_v]),
                          np.concatenate([mesh.normals, vertex_normals(pane_v, pane_f)]),
                          np.concatenate([mesh.indices, pane_f + np.uint32(nv)]),
                          np.concatenate([tm, np.full(len(pane_f), base + 3, dtype=np.uint32)]))]
    return sc, cam_m, yfov


def glass_scene(view="room"):
    """The box of box_scene with glass and add materials (view="pane": the camera looks up at the area light
    through the pane, so most glass vertices are re-sampled ones whose transmitted ray runs into the light): a solid glass sphere (absorbing), a sphere that is
    half diffuse / half glass (constant mix, as a glTF transmission factor builds it) under a faint emission
    (add material), a glass pane right below the area light (seen from behind: internal rays, total internal
    reflection at grazing angles, transmitted rays that run into the light), a glowing left wall (emission added
    to green diffuse) and a floor whose emission is added to a diffuse / rough glossy mix."""
    sc, cam_m, yfov = box_scene()
    base = len(sc.materials)
    sc.materials += [
        MaterialData(MAT_GLASS, (0.75, 0.95, 0.85), intensity=1.5, roughness=2.0),   # base+0: ior 1.5, attenuation 2
        MaterialData(MAT_CONSTMIX, mix_factor=0.5, first=0, second=base + 0),         # base+1: transmission 0.5
        MaterialData(MAT_EMISSIVE, (0.05, 0.1, 0.3), intensity=1.0),                  # base+2
        MaterialData(MAT_ADD, first=base + 2, second=base + 1),                       # base+3: emission + (diffuse | glass)
        MaterialData(MAT_GLASS, (1.0, 1.0, 1.0), intensity=1.33, roughness=5.0),      # base+4: clear pane
        MaterialData(MAT_ADD, first=base + 2, second=3),                              # base+5: emission + green diffuse
        MaterialData(MAT_CONSTMIX, mix_factor=0.4, first=0, second=1),                # base+6: diffuse | rough glossy
        MaterialData(MAT_EMISSIVE, (0.2, 0.05, 0.02), intensity=0.5),                 # base+7
        MaterialData(MAT_ADD, first=base + 7, second=base + 6),                       # base+8: emission + mix
    ]
    mesh = sc.meshes[0]
    tm = mesh.tri_material.copy()
    n = 6
    quad = 2 * n * n
    tm[0:quad] = base + 8                      # floor
    tm[4 * quad:5 * quad] = base + 5           # left wall
    nsph = (len(tm) - 6 * quad) // 2
    tm[6 * quad:6 * quad + nsph] = base + 0    # first sphere: glass
    tm[6 * quad + nsph:] = base + 3            # second sphere
    pane_v, pane_f = grid_quad((1.0, 3.0, 0.5), (2.0, 0.0, 0.0), (0.0, 0.0, 2.0), 2, 2)
    pane_v = np.asarray(pane_v, dtype=np.float32)
    pane_f = np.asarray(pane_f, dtype=np.uint32)
    nv = len(mesh.vertices)
    sc.meshes = [MeshData(np.concatenate([mesh.vertices, pane_v]),
                          np.concatenate([mesh.normals, vertex_normals(pane_v, pane_f)]),
                          np.concatenate([mesh.indices, pane_f + np.uint32(nv)]),
                          np.concatenate([tm, np.full(len(pane_f), base + 4, dtype=np.uint32)]))]
    if view == "pane":
        return sc, look_at((2.2, 0.3, 2.9), (2.0, 3.95, 1.8), up=(0.0, 0.0, -1.0)), 0.7
    return sc, look_at((2.0, 1.0, 3.9), (2.0, 2.4, 0.5)), yfov


def sun_scene(lights="sun+point"):
    """Open scene under a directional light (light/DirectionalLight.h): a diffuse floor, a diffuse and a
    rough glossy sphere, a sky-coloured environment.  lights: "sun" (directional only), "sun+point"
    (two light types), "all" (point + area + two directional lights), "sky-map" (no lamp at all: an image-map
    environment lights the scene) or "sky" (the same under the SkyEnvironment gradient)."""
    parts = [(*grid_quad((-4, 0, -4), (0, 0, 8), (8, 0, 0), 4, 4), 0)]
    us = uv_sphere(16, 8)
    parts.append((us.vertices.astype(np.float64) * 0.8 + np.array([-1.0, 0.8, 0.3]), us.indices, 1))
    parts.append((us.vertices.astype(np.float64) * 0.6 + np.array([1.1, 0.6, -0.4]), us.indices, 2))
    sc = SceneData()
    sc.meshes = [_merge(parts)]
    sc.materials = [MaterialData(MAT_DIFFUSE, (0.6, 0.6, 0.55)), MaterialData(MAT_DIFFUSE, (0.7, 0.25, 0.2)),
                    MaterialData(MAT_GLOSSY, (0.9, 0.9, 0.9), roughness=0.3)]
    sc.instances = [InstanceData(0, 0, np.eye(4), np.eye(4))]
    d = np.array([-0.35, -1.0, -0.25]); d /= np.linalg.norm(d)
    sc.directional_lights = [(tuple(d), 0.05, (1.0, 0.97, 0.9), 2.5)]
    if lights in ("sun+point", "all"):
        sc.point_lights = [((2.5, 3.0, 2.0), (1.0, 1.0, 1.0), 60.0)]
    if lights == "all":
        sc.area_lights = [((-1.0, 3.5, -1.0), (1.0, 3.5, -1.0), (0.0, 3.5, 1.0), (1.0, 0.9, 0.8), 80.0)]
        d2 = np.array([0.6, -0.7, 0.4]); d2 /= np.linalg.norm(d2)
        sc.directional_lights.append((tuple(d2), 0.0, (0.3, 0.4, 1.0), 0.8))
    if lights == "sky":
        sc.sky = True            # SkyEnvironment lights the scene, no lamp at all
        sc.directional_lights = []
    elif lights == "sky-map":
        # an image-map environment: smooth sky gradient with a few very bright texels (a sun) and random detail
        rng = np.random.RandomState(8)
        h, w = 32, 64
        texels = np.ones((h, w, 4), np.float32)
        texels[..., :3] = (np.linspace(1.2, 0.1, h, dtype=np.float32)[:, None, None] * np.array([0.5, 0.7, 1.0], np.float32)
                           + rng.uniform(0.0, 0.2, size=(h, w, 3)).astype(np.float32))
        texels[5:7, 40:43, :3] = (60.0, 55.0, 40.0)
        sc.env_map = (texels, 0.8)
        sc.directional_lights = []
    else:
        sc.env = ((0.25, 0.35, 0.5), 1.0)
    if lights == "sky":
        sc.env = None
    cam_m = look_at((0.3, 2.2, 5.0), (0.0, 0.6, 0.0))
    return sc, cam_m, 0.8


def textured_scene(seed=3, normal_maps=False, plain_mesh=False, drop_texcoords=False):
    """The box of box_scene with an albedo map on the floor and the back wall (texture coordinates beyond
    [0, 1) and negative ones included), a gamma-corrected map on one sphere and an emission map on a small panel.
    normal_maps: a bump pattern perturbs the shading normal of the floor (diffuse, with its albedo map), of the
    first sphere (rough glossy) and of the right wall (mirror).
    plain_mesh: a second mesh without texture coordinates (a scaled, rotated sphere instance) wears the floor's
    material.  drop_texcoords: no mesh has texture coordinates (every lookup reads texel (0, 0), tangents are b - a)."""
    rng = np.random.RandomState(seed)
    sc, cam_m, yfov = box_scene()
    mesh = sc.meshes[0]
    v = mesh.vertices
    uv = np.stack([v[:, 0] * 0.55 - 0.3 + 0.35 * v[:, 1], v[:, 2] * 0.6 - 0.4 + 0.2 * v[:, 1]], axis=1).astype(np.float32)
    panel_v, panel_f = grid_quad((1.2, 2.2, 0.05), (1.6, 0.0, 0.0), (0.0, 1.0, 0.0), 2, 2)
    panel_v = np.asarray(panel_v, dtype=np.float32)
    panel_f = np.asarray(panel_f, dtype=np.uint32)
    panel_uv = np.stack([(panel_v[:, 0] - 1.2) / 1.6, (panel_v[:, 1] - 2.2) / 1.0], axis=1).astype(np.float32)
    base = len(sc.materials)
    checker = rng.randint(40, 256, size=(8, 8, 4)).astype(np.uint8)
    stripes = np.zeros((4, 16, 4), np.uint8)
    stripes[:, ::2] = (250, 220, 90, 255)
    stripes[:, 1::2] = (40, 60, 200, 255)
    glow = rng.randint(0, 256, size=(3, 5, 4)).astype(np.uint8)
    sc.textures = [(checker, 1.0), (stripes, 2.2), (glow, 1.0)]
    bump = None
    if normal_maps:
        bumps = np.empty((16, 16, 4), np.uint8)
        bumps[..., 0] = rng.randint(70, 186, size=(16, 16))
        bumps[..., 1] = rng.randint(70, 186, size=(16, 16))
        bumps[..., 2] = rng.randint(200, 256, size=(16, 16))
        bumps[..., 3] = 255
        sc.textures.append((bumps, 1.0))
        bump = 3
    sc.materials += [MaterialData(MAT_DIFFUSE, (1.0, 1.0, 1.0), texture=0, normal_map=bump), MaterialData(MAT_DIFFUSE, (1.0, 1.0, 1.0), texture=1),
                     MaterialData(MAT_EMISSIVE, (1.0, 1.0, 1.0), intensity=1.5, texture=2),
                     MaterialData(MAT_GLOSSY, (0.9, 0.9, 0.9), roughness=0.25, normal_map=bump),
                     MaterialData(MAT_GLOSSY, (0.95, 0.95, 0.95), roughness=0.0, normal_map=bump)]
    tm = mesh.tri_material.copy()
    quad = 2 * 6 * 6
    tm[0:quad] = base            # floor
    tm[2 * quad:3 * quad] = base  # back wall
    nsph = (len(tm) - 6 * quad) // 2
    tm[6 * quad + nsph:] = base + 1   # second sphere
    if normal_maps:
        tm[6 * quad:6 * quad + nsph] = base + 3   # first sphere
        tm[5 * quad:6 * quad] = base + 4          # right wall
    nv = len(mesh.vertices)
    sc.meshes = [MeshData(np.concatenate([mesh.vertices, panel_v]), np.concatenate([mesh.normals, vertex_normals(panel_v, panel_f)]),
                          np.concatenate([mesh.indices, panel_f + np.uint32(nv)]),
                          np.concatenate([tm, np.full(len(panel_f), base + 2, dtype=np.uint32)]),
                          np.concatenate([uv, panel_uv]))]
    if plain_mesh:
        us = uv_sphere(12, 6)
        sc.meshes.append(MeshData(us.vertices, us.normals, us.indices))
        m, inv = trs((2.1, 1.9, 1.5), (0.0, 0.38268343, 0.0, 0.92387953), (0.45, 0.3, 0.45))
        sc.instances.append(InstanceData(1, base, m, inv))
    if drop_texcoords:
        for k, me in enumerate(sc.meshes):
            sc.meshes[k] = MeshData(me.vertices, me.normals, me.indices, me.tri_material)
    return sc, cam_m, yfov


def random_rays(n, lo, hi, seed=0):
    """Incoherent rays: origins uniform in the box [lo,hi], directions uniform on the sphere."""
    rng = np.random.RandomState(seed)
    o = rng.uniform(np.asarray(lo, dtype=np.float64), np.asarray(hi, dtype=np.float64), size=(n, 3)).astype(np.float32)
    d = rng.normal(size=(n, 3))
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    return o, d.astype(np.float32)
