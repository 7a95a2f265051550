"""The .glb -> rtgpu_scene_desc bridge (host code of the library; no GPU needed).

What loadGLTFScene + DynamicScene::build / soupifyScene produce for a file (raytracer/src/lib/io/GLTF.cpp,
scene/dynamic/DynamicScene.cpp) is checked field by field on files written by `gltf.GlbBuilder`: the reference
ships no glTF assets and its loader (tinygltf) is not built here, so the expected values are derived from the
loader's source, cited per assertion.
"""
import os
import struct

import numpy as np
import pytest

import oracle
from raytracer_amd import abi, api, gltf, scenes


def _quad(z=0.0):
    v = np.array([[-1, -1, z], [1, -1, z], [1, 1, z], [-1, 1, z]], np.float32)
    n = np.tile(np.array([[0, 0, 1]], np.float32), (4, 1))
    uv = np.array([[0, 0], [1, 0], [1, 1], [0, 1]], np.float32)
    return v, n, np.array([[0, 1, 2], [0, 2, 3]], np.uint32), uv


def _full_scene(tmp_path):
    b = gltf.GlbBuilder()
    rng = np.random.RandomState(0)
    albedo = rng.randint(0, 256, size=(4, 8, 4)).astype(np.uint8)
    bumps = rng.randint(100, 156, size=(2, 2, 4)).astype(np.uint8)
    t_albedo, t_bumps = b.add_texture(albedo), b.add_texture(bumps)
    m_plain = b.add_material(pbrMetallicRoughness={"baseColorFactor": [0.5, 0.25, 0.125, 1.0]})
    m_rich = b.add_material(pbrMetallicRoughness={"baseColorFactor": [0.9, 0.8, 0.7, 1.0], "metallicFactor": 0.25, "roughnessFactor": 0.3,
                                                  "baseColorTexture": {"index": t_albedo}},
                            normalTexture={"index": t_bumps}, emissiveFactor=[0.1, 0.2, 0.3])
    v, n, idx, uv = _quad()
    us = scenes.uv_sphere(8, 4)
    mesh_quad = b.add_primitive_mesh([(v, n, idx, uv, m_rich)])
    mesh_two = b.add_primitive_mesh([(us.vertices, us.normals, us.indices, None, m_plain), (v, n, idx, None, None)])
    root = b.add_node(translation=[1.0, 2.0, 3.0], rotation=[0.0, 0.38268343, 0.0, 0.92387953], scale=[2.0, 2.0, 2.0],
                      extras={"IsMainCamera": 1, "FocalDistance": 4.0, "FStop": 2.8})
    b.add_node(parent=root, mesh=mesh_quad, translation=[0.0, 0.5, 0.0],
               extras={"Material.ClearCoat": True, "Material.ClearCoatIOR": 1.6, "Material.ClearCoatRoughness": 0.05,
                       "Material.Transmission": 0.4, "Material.IOR": 1.33, "Material.AttenuationStrength": 3,
                       "Material.EmissionStrength": 2.5, "Material.Alpha": 0.75})
    b.add_node(mesh=mesh_quad, translation=[5.0, 0.0, 0.0])      # the same mesh again: one shape, two instances
    b.add_node(mesh=mesh_two, matrix=[1, 0, 0, 0, 0, 0, 1, 0, 0, -1, 0, 0, -3, 0, 1, 1])   # column-major: rotation about x + translation
    b.add_node(parent=root, camera=b.add_camera(0.6))
    b.add_node(camera=b.add_camera(0.9), translation=[0.0, 0.0, 9.0])
    b.add_node(translation=[0.0, 4.0, 0.0], extras={"IsPointLight": True, "LightIntensity": 120.0, "LightColor": [1.0, 0.5, 0.25]})
    b.add_node(extras={"IsPointLight": True, "LightIntensity": 0.0})   # dropped: intensity < 1e-6 (io/GLTF.cpp:547-550)
    tri_v = np.array([[0, 0, 0], [1, 0, 0], [0, 0, 1]], np.float32)
    tri = b.add_primitive_mesh([(tri_v, np.tile(np.array([[0, -1, 0]], np.float32), (3, 1)), np.array([[0, 1, 2]]), None, None)])
    b.add_node(mesh=tri, translation=[0.0, 6.0, 0.0], scale=[2.0, 1.0, 2.0], extras={"IsAreaLight": True})
    b.add_node(rotation=[0.70710678, 0.0, 0.0, 0.70710678], extras={"IsDirectionalLight": True, "LightIntensity": 3, "DirectionalLight.Angle": 0.02})
    b.set_scene_extras(EnvironmentIntensity=0.5, EnvironmentColor=[0.2, 0.3, 0.4])
    path = os.path.join(tmp_path, "scene.glb")
    b.write(path)
    return path, albedo, bumps


def test_nodes_materials_lights_and_camera_of_a_file(tmp_path):
    path, albedo, bumps = _full_scene(str(tmp_path))
    with gltf.load_glb(path, image_aspect_ratio=16 / 9) as g:
        sc = g.scene_data()
        cam = g.camera()
        assert g.camera_count() == 2
    # ---- instances: one per (node, primitive), transforms accumulated parent * child (DynamicScene.cpp:118) ----
    assert len(sc.meshes) == 3 and [i.shape for i in sc.instances] == [0, 0, 1, 2]   # meshCache: the quad is stored once
    root_m, root_inv = scenes.trs((1.0, 2.0, 3.0), (0.0, 0.38268343, 0.0, 0.92387953), (2.0, 2.0, 2.0))
    want = root_m @ scenes.translate(0.0, 0.5, 0.0)
    assert np.allclose(sc.instances[0].transform, want, atol=1e-12)
    assert np.allclose(sc.instances[0].inverse @ want, np.eye(4), atol=1e-12)
    assert np.allclose(sc.instances[1].transform, scenes.translate(5.0, 0.0, 0.0))
    col_major = np.array([1, 0, 0, 0, 0, 0, 1, 0, 0, -1, 0, 0, -3, 0, 1, 1], float).reshape(4, 4).T
    assert np.allclose(sc.instances[2].transform, col_major) and np.allclose(sc.instances[3].transform, col_major)
    assert np.allclose(sc.instances[2].inverse @ col_major, np.eye(4), atol=1e-12)
    assert sc.meshes[0].texcoords is not None and sc.meshes[1].texcoords is None
    assert sc.meshes[1].indices.shape == (scenes.uv_sphere(8, 4).indices.shape[0], 3)

    # ---- the material tree of loadMaterial (io/GLTF.cpp:323-455), outermost first ----
    M = sc.materials
    top = M[sc.instances[0].material]
    assert top.type == abi.MAT_CONSTMIX and top.mix_factor == 0.75 and M[top.first].type == abi.MAT_TRANSPARENT          # :444-452 alpha
    add = M[top.second]
    assert add.type == abi.MAT_ADD                                                                                      # :435-441
    em = M[add.first]
    assert em.type == abi.MAT_EMISSIVE and np.allclose(em.color, (0.1, 0.2, 0.3)) and em.intensity == 2.5               # :426-433
    trans = M[add.second]
    assert trans.type == abi.MAT_CONSTMIX and trans.mix_factor == 0.4                                                   # :399-410
    glass = M[trans.second]
    assert glass.type == abi.MAT_GLASS and glass.intensity == 1.33 and glass.roughness == 3.0 and np.allclose(glass.color, (0.9, 0.8, 0.7))
    coat = M[trans.first]
    assert coat.type == abi.MAT_FRESNELMIX and np.isclose(coat.roughness, 1.6)                                          # :349-360
    coat_glossy = M[coat.second]
    assert coat_glossy.type == abi.MAT_GLOSSY and np.isclose(coat_glossy.roughness, 0.05) and coat_glossy.normal_map is not None
    metal = M[coat.first]
    assert metal.type == abi.MAT_CONSTMIX and metal.mix_factor == 0.25                                                  # :335-346
    diffuse, glossy = M[metal.first], M[metal.second]
    assert diffuse.type == abi.MAT_DIFFUSE and np.allclose(diffuse.color, (0.9, 0.8, 0.7)) and diffuse.intensity == 1.0
    assert glossy.type == abi.MAT_GLOSSY and np.isclose(glossy.roughness, 0.3) and np.allclose(glossy.color, (0.9, 0.8, 0.7))
    assert diffuse.texture is not None and diffuse.normal_map == glossy.normal_map == coat_glossy.normal_map
    px, gamma = sc.textures[diffuse.texture]
    assert gamma == 2.2 and np.array_equal(px, albedo)                                                                  # :362-369
    px, gamma = sc.textures[diffuse.normal_map]
    assert gamma == 1.0 and np.array_equal(px, bumps)
    # materialCache (:473-493): the second user of the material gets the tree built with the first node's extras
    assert sc.instances[1].material == sc.instances[0].material
    plain = M[sc.instances[2].material]
    assert plain.type == abi.MAT_DIFFUSE and np.allclose(plain.color, (0.5, 0.25, 0.125)) and plain.texture is None
    dflt = M[sc.instances[3].material]
    assert dflt.type == abi.MAT_DIFFUSE and np.allclose(dflt.color, (1, 1, 1)) and dflt.intensity == 0.8                # :481-486

    # ---- lights (io/GLTF.cpp:517-561, applyTransform of light/*.cpp) ----
    assert len(sc.point_lights) == 1
    pos, col, inten = sc.point_lights[0]
    assert np.allclose(pos, (0, 4, 0)) and np.allclose(col, (1.0, 0.5, 0.25)) and inten == 120.0
    (a, bb, c, col, inten), = sc.area_lights
    assert np.allclose(a, (0, 6, 0)) and np.allclose(bb, (2, 6, 0)) and np.allclose(c, (0, 6, 2)) and inten == 500 and np.allclose(col, 1)
    (d, angle, col, inten), = sc.directional_lights
    assert np.allclose(d, (0, 0, -1), atol=1e-6) and np.isclose(angle, 0.01) and inten == 3.0   # (0,-1,0) turned 90 degrees about x
    assert sc.env is not None and np.allclose(sc.env[0], (0.2, 0.3, 0.4)) and sc.env[1] == 0.5   # :650-680

    # ---- camera: the one under the IsMainCamera node wins over the first-listed... both are listed; main is the child ----
    x_fov = np.degrees(2 * np.arctan(np.tan(0.6 / 2) * 16 / 9))                                 # :585-587
    ref = api.camera_from_transform(x_fov, root_m)
    for f in ("origin", "u", "v", "w"):
        assert np.allclose(getattr(cam, f)[:], getattr(ref, f)[:], atol=1e-6)
    assert np.isclose(cam.width, ref.width) and cam.focal_distance == 4.0
    dist_to_sensor = 0.5 * 0.036 / np.tan(0.5 * np.radians(x_fov))
    assert np.isclose(cam.aperture, 0.5 * (1.0 / (1.0 / 4.0 + 1.0 / dist_to_sensor)) / 2.8)       # :612-625 (FStop on the parent)


def test_soupify_merges_instances_in_world_space(tmp_path):
    """--soupify (DynamicScene::soupifyScene, scene/dynamic/DynamicScene.cpp:17-106): one mesh, vertices moved to
    world space, per-triangle materials; rays find the same surfaces at the same distances as in the instanced scene."""
    path, _, _ = _full_scene(str(tmp_path))
    with gltf.load_glb(path, 1.0) as g, gltf.load_glb(path, 1.0, soupify=True) as s:
        inst, soup = g.scene_data(), s.scene_data()
    assert len(soup.meshes) == 1 and len(soup.instances) == 1 and np.allclose(soup.instances[0].transform, np.eye(4))
    n_tris = [inst.meshes[i.shape].indices.shape[0] for i in inst.instances]
    assert soup.meshes[0].indices.shape[0] == sum(n_tris)
    want_mat = np.concatenate([np.full(n, i.material, np.uint32) for n, i in zip(n_tris, inst.instances)])
    assert np.array_equal(soup.meshes[0].tri_material, want_mat)                                  # CompositeMaterial ranges, :67-71
    v0 = inst.meshes[0].vertices
    moved = (np.asarray(inst.instances[0].transform) @ np.c_[v0, np.ones(len(v0))].T).T[:, :3]
    assert np.allclose(soup.meshes[0].vertices[:len(v0)], moved, atol=1e-5)
    o, d = scenes.random_rays(20000, (-6, -3, -6), (8, 6, 8), seed=5)
    a = oracle.OracleScene(inst).trace_closest(o, d)
    b = oracle.OracleScene(soup).trace_closest(o, d)
    hit = a["inst"] != abi.INVALID_ID
    assert hit.sum() > 200 and np.array_equal(hit, b["inst"] != abi.INVALID_ID)
    assert np.allclose(a["t"][hit], b["t"][hit], rtol=2e-4)


def test_loader_errors(tmp_path):
    p = str(tmp_path)
    with pytest.raises(api.RtgpuError, match="cannot open"):
        gltf.load_glb(os.path.join(p, "missing.glb"))
    bad = os.path.join(p, "bad.glb")
    open(bad, "wb").write(b"not a glb file at all....")
    with pytest.raises(api.RtgpuError, match="rtgpu error 7"):
        gltf.load_glb(bad)
    # no camera: Renderer::findCamera throws (renderer/Renderer.cpp:316-318)
    b = gltf.GlbBuilder()
    v, n, idx, uv = _quad()
    b.add_node(mesh=b.add_primitive_mesh([(v, n, idx, None, None)]))
    b.write(os.path.join(p, "nocam.glb"))
    with gltf.load_glb(os.path.join(p, "nocam.glb")) as g:
        with pytest.raises(api.RtgpuError, match="No camera in scene."):
            g.camera()
    # implicit normals (io/GLTF.cpp:231-234)
    b = gltf.GlbBuilder()
    m = b.add_primitive_mesh([(v, n, idx, None, None)])
    del b.doc["meshes"][m]["primitives"][0]["attributes"]["NORMAL"]
    b.add_node(mesh=m)
    b.write(os.path.join(p, "nonormals.glb"))
    with pytest.raises(api.RtgpuError, match="Implicit normals are not supported"):
        gltf.load_glb(os.path.join(p, "nonormals.glb"))
    # an area light must be one triangle (:533-536)
    b = gltf.GlbBuilder()
    b.add_node(mesh=b.add_primitive_mesh([(v, n, idx, None, None)]), extras={"IsAreaLight": True})
    b.write(os.path.join(p, "arealight.glb"))
    with pytest.raises(api.RtgpuError, match="must be triangular"):
        gltf.load_glb(os.path.join(p, "arealight.glb"))
    # JPEG data has no decoder here
    b = gltf.GlbBuilder()
    b.doc["images"].append({"bufferView": b._view(b"\xff\xd8\xff\xe0" + b"\x00" * 16), "mimeType": "image/jpeg"})
    b.doc["textures"].append({"source": 0})
    mat = b.add_material(pbrMetallicRoughness={"baseColorTexture": {"index": 0}})
    b.add_node(mesh=b.add_primitive_mesh([(v, n, idx, uv, mat)]))
    b.write(os.path.join(p, "jpeg.glb"))
    with pytest.raises(api.RtgpuError, match="rtgpu error 8"):
        gltf.load_glb(os.path.join(p, "jpeg.glb"))
    # a truncated chunk
    data = open(os.path.join(p, "nocam.glb"), "rb").read()
    open(os.path.join(p, "cut.glb"), "wb").write(data[:40] + struct.pack("<I", 1 << 30) + data[44:60])
    with pytest.raises(api.RtgpuError):
        gltf.load_glb(os.path.join(p, "cut.glb"))


def test_png_round_trip_with_filters(tmp_path):
    """The PNG reader against zlib-compressed files with every row filter type (written by hand below)."""
    import zlib
    rng = np.random.RandomState(3)
    px = rng.randint(0, 256, size=(5, 7, 4)).astype(np.uint8)
    rows = []
    prev = np.zeros(7 * 4, np.int32)
    for y in range(5):
        cur = px[y].reshape(-1).astype(np.int32)
        f = y % 5
        left = np.concatenate([np.zeros(4, np.int32), cur[:-4]])
        upleft = np.concatenate([np.zeros(4, np.int32), prev[:-4]])
        if f == 0: pred = np.zeros_like(cur)
        elif f == 1: pred = left
        elif f == 2: pred = prev
        elif f == 3: pred = (left + prev) // 2
        else:
            p = left + prev - upleft
            pa, pb, pc = np.abs(p - left), np.abs(p - prev), np.abs(p - upleft)
            pred = np.where((pa <= pb) & (pa <= pc), left, np.where(pb <= pc, prev, upleft))
        rows.append(bytes([f]) + ((cur - pred) % 256).astype(np.uint8).tobytes())
        prev = cur

    def chunk(tag, body):
        return struct.pack(">I", len(body)) + tag + body + struct.pack(">I", zlib.crc32(tag + body) & 0xFFFFFFFF)
    comp = zlib.compress(b"".join(rows))
    png = (b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", 7, 5, 8, 6, 0, 0, 0)) + chunk(b"IDAT", comp[:20]) + chunk(b"IDAT", comp[20:])
           + chunk(b"IEND", b""))
    b = gltf.GlbBuilder()
    b.doc["images"].append({"bufferView": b._view(png), "mimeType": "image/png"})
    b.doc["textures"].append({"source": 0})
    v, n, idx, uv = _quad()
    mat = b.add_material(pbrMetallicRoughness={"baseColorTexture": {"index": 0}})
    b.add_node(mesh=b.add_primitive_mesh([(v, n, idx, uv, mat)]))
    path = os.path.join(str(tmp_path), "png.glb")
    b.write(path)
    with gltf.load_glb(path) as g:
        (got, gamma), = g.scene_data().textures
    assert np.array_equal(got, px) and gamma == 2.2


def _write_png(samples, depth, ctype, interlace=False, palette=None, trns=None, filters=(0, 1, 2, 3, 4)):
    """A PNG file of `samples` (height, width, channels) of `depth` bits each, by the PNG specification: rows packed
    most significant bit first, every row filter type in turn, optionally the seven Adam7 passes."""
    import zlib
    h, w, ch = samples.shape
    bits = ch * depth
    bpp = max(1, bits // 8)

    def pack_rows(img):
        rows = []
        for y in range(img.shape[0]):
            if depth == 8:
                rows.append(img[y].astype(np.uint8).reshape(-1))
            elif depth == 16:
                rows.append(img[y].astype(">u2").reshape(-1).view(np.uint8))
            else:
                v = img[y].reshape(-1)
                bitsv = ((v[:, None] >> np.arange(depth - 1, -1, -1)) & 1).astype(np.uint8).reshape(-1)
                rows.append(np.packbits(bitsv))
        return rows

    def filtered(rows, first_filter):
        out = []
        prev = np.zeros(len(rows[0]), np.int32) if rows else None
        for y, r in enumerate(rows):
            cur = r.astype(np.int32)
            f = filters[(first_filter + y) % len(filters)]
            left = np.concatenate([np.zeros(bpp, np.int32), cur[:-bpp]]) if len(cur) > bpp else np.zeros_like(cur)
            upleft = np.concatenate([np.zeros(bpp, np.int32), prev[:-bpp]]) if len(cur) > bpp else np.zeros_like(cur)
            if f == 0: pred = np.zeros_like(cur)
            elif f == 1: pred = left
            elif f == 2: pred = prev
            elif f == 3: pred = (left + prev) // 2
            else:
                q = left + prev - upleft
                pa, pb, pc = np.abs(q - left), np.abs(q - prev), np.abs(q - upleft)
                pred = np.where((pa <= pb) & (pa <= pc), left, np.where(pb <= pc, prev, upleft))
            out.append(bytes([f]) + ((cur - pred) % 256).astype(np.uint8).tobytes())
            prev = cur
        return b"".join(out)

    if interlace:
        raw = b""
        for k, (x0, y0, dx, dy) in enumerate(((0, 0, 8, 8), (4, 0, 8, 8), (0, 4, 4, 8), (2, 0, 4, 4), (0, 2, 2, 4), (1, 0, 2, 2), (0, 1, 1, 2))):
            sub = samples[y0::dy, x0::dx]
            if sub.shape[0] and sub.shape[1]:
                raw += filtered(pack_rows(sub), k)
    else:
        raw = filtered(pack_rows(samples), 0)

    def chunk(tag, body):
        return struct.pack(">I", len(body)) + tag + body + struct.pack(">I", zlib.crc32(tag + body) & 0xFFFFFFFF)
    comp = zlib.compress(raw)
    out = b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, depth, ctype, 0, 0, 1 if interlace else 0))
    if palette is not None:
        out += chunk(b"PLTE", np.asarray(palette, np.uint8).tobytes())
    if trns is not None:
        out += chunk(b"tRNS", bytes(trns))
    cut = len(comp) // 3
    return out + chunk(b"IDAT", comp[:cut]) + chunk(b"IDAT", comp[cut:]) + chunk(b"IEND", b"")


def _texture_of(png, tmp_path, name="img.glb"):
    b = gltf.GlbBuilder()
    b.doc["images"].append({"bufferView": b._view(png), "mimeType": "image/png"})
    b.doc["textures"].append({"source": 0})
    v, n, idx, uv = _quad()
    mat = b.add_material(pbrMetallicRoughness={"baseColorTexture": {"index": 0}})
    b.add_node(mesh=b.add_primitive_mesh([(v, n, idx, uv, mat)]))
    path = os.path.join(str(tmp_path), name)
    b.write(path)
    with gltf.load_glb(path) as g:
        (got, _), = g.scene_data().textures
    return got


@pytest.mark.parametrize("interlace", [False, True])
def test_png_colour_types_bit_depths_and_interlacing(tmp_path, interlace):
    """Every colour type and bit depth a PNG can have, plain and Adam7, written from known sample values by the PNG
    specification; expected RGBA by stb_image's conventions, which the reference's loader inherits (read from
    raytracer/src/lib/io/lib/stb_image.h: stbi__depth_scale_table -- grey of 1 / 2 / 4 bits times 0xff / 0x55 / 0x11,
    palette indices unscaled; stbi__compute_transparency -- a tRNS colour key gives alpha 0 / 255; per-entry alpha for
    palettes).  The reference's decoder could not be run here: parity unpinned, the test pins these rules.
    Sizes 1..9 pixels a side cover Adam7 passes that are empty."""
    rng = np.random.RandomState(5)
    for (w, h) in ((9, 7), (1, 1), (3, 2), (8, 8), (5, 9)):
        for depth in (1, 2, 4, 8):
            top = (1 << depth) - 1
            scale = {1: 0xFF, 2: 0x55, 4: 0x11, 8: 1}[depth]
            # grey, with and without a colour key
            smp = rng.randint(0, top + 1, size=(h, w, 1))
            for key in (None, int(smp[0, 0, 0])):
                got = _texture_of(_write_png(smp, depth, 0, interlace, trns=None if key is None else struct.pack(">H", key)), tmp_path)
                grey = (smp[..., 0] * scale).astype(np.uint8)
                alpha = np.full((h, w), 255, np.uint8) if key is None else np.where(smp[..., 0] == key, 0, 255).astype(np.uint8)
                assert np.array_equal(got, np.stack([grey, grey, grey, alpha], axis=-1)), (w, h, depth, key)
            # palette, with per-entry alpha for the first entries
            pal = rng.randint(0, 256, size=(top + 1, 3)).astype(np.uint8)
            tr = rng.randint(0, 256, size=(top + 1) // 2 + 1).astype(np.uint8)
            idx = rng.randint(0, top + 1, size=(h, w, 1))
            got = _texture_of(_write_png(idx, depth, 3, interlace, palette=pal, trns=tr.tobytes()), tmp_path)
            alpha = np.where(idx[..., 0] < len(tr), tr[np.minimum(idx[..., 0], len(tr) - 1)], 255).astype(np.uint8)
            assert np.array_equal(got, np.concatenate([pal[idx[..., 0]], alpha[..., None]], axis=-1)), (w, h, depth, "palette")
        # 8-bit RGB (with a colour key), grey + alpha, RGBA
        rgb = rng.randint(0, 256, size=(h, w, 3))
        got = _texture_of(_write_png(rgb, 8, 2, interlace, trns=struct.pack(">HHH", *[int(c) for c in rgb[-1, -1]])), tmp_path)
        alpha = np.where((rgb == rgb[-1, -1]).all(-1), 0, 255)
        assert np.array_equal(got, np.concatenate([rgb, alpha[..., None]], axis=-1).astype(np.uint8))
        ga = rng.randint(0, 256, size=(h, w, 2))
        got = _texture_of(_write_png(ga, 8, 4, interlace), tmp_path)
        assert np.array_equal(got, np.stack([ga[..., 0]] * 3 + [ga[..., 1]], axis=-1).astype(np.uint8))
        rgba = rng.randint(0, 256, size=(h, w, 4))
        assert np.array_equal(_texture_of(_write_png(rgba, 8, 6, interlace), tmp_path), rgba.astype(np.uint8))


def test_png_files_the_reference_refuses(tmp_path):
    """16-bit images reach loadImage with 8 bytes per pixel and fail its size check ("Can't load image data",
    io/GLTF.cpp:311-313); malformed headers fail in the decoder (stb_image: "Corrupt PNG")."""
    rng = np.random.RandomState(2)
    for ctype, ch in ((0, 1), (2, 3), (4, 2), (6, 4)):
        with pytest.raises(api.RtgpuError, match="Can't load image data"):
            _texture_of(_write_png(rng.randint(0, 65536, size=(3, 4, ch)), 16, ctype), tmp_path)
    bad = [(_write_png(rng.randint(0, 4, size=(3, 4, 3)), 2, 2), "colour type"),                                   # RGB needs 8 or 16 bits
           (_write_png(rng.randint(0, 256, size=(3, 4, 4)), 8, 6, trns=b"\x00\x01"), "tRNS"),                       # tRNS beside an alpha channel
           (_write_png(rng.randint(0, 256, size=(3, 4, 3)), 8, 2, trns=b"\x00\x01"), "tRNS"),                       # wrong key length
           (_write_png(rng.randint(0, 4, size=(3, 4, 1)), 2, 3), "palette"),                                       # no PLTE
           (_write_png(rng.randint(0, 4, size=(3, 4, 1)), 2, 3, palette=np.zeros((2, 3), np.uint8), trns=b"\x01\x02\x03"), "tRNS")]
    for png, what in bad:
        with pytest.raises(api.RtgpuError, match=what):
            _texture_of(png, tmp_path)


def _env_scene(tmp_path, hdr_bytes, hdr_name="#WORKDIR#/sky.hdr", intensity=1.5):
    open(os.path.join(str(tmp_path), "sky.hdr"), "wb").write(hdr_bytes)
    b = gltf.GlbBuilder()
    us = scenes.uv_sphere(12, 6)
    b.add_node(mesh=b.add_primitive_mesh([(us.vertices, us.normals, us.indices, None, None)]))
    cam_m = scenes.look_at((0.0, 0.5, 4.0), (0.0, 0.0, 0.0))
    b.add_node(camera=b.add_camera(0.8), matrix=[float(x) for x in np.asarray(cam_m).T.reshape(-1)])
    b.set_scene_extras(EnvironmentIntensity=intensity, EnvironmentHDRIFilePath=hdr_name, EnvironmentColor=[1, 0, 0])
    path = os.path.join(str(tmp_path), "env.glb")
    b.write(path)
    return path


def test_hdr_environment_map_of_a_file(tmp_path):
    """loadEnvironmentInfo with "EnvironmentHDRIFilePath" (io/GLTF.cpp:640-660): the Radiance file is found through the
    work directory ("#WORKDIR#", utility/PathResolver.cpp) and read as stb_image reads it for read_hdr_file
    (io/HDR.cpp:5-18; rules read from io/lib/stb_image.h stbi__hdr_load / stbi__hdr_convert: mantissa * 2^(e - 136),
    black for e = 0, alpha 1; run-length coded rows for 8 <= width < 32768, flat rows otherwise).  The HDR file wins
    over "EnvironmentColor".  The reference's decoder could not be run here: parity unpinned, the test pins the rules."""
    rng = np.random.RandomState(4)
    gltf.set_workdir(str(tmp_path))
    try:
        for (w, h, coded) in ((16, 6, True), (16, 6, False), (5, 3, True), (40, 9, True)):
            rgbe = rng.randint(0, 256, size=(h, w, 4)).astype(np.uint8)
            rgbe[..., 3] = rng.randint(120, 140, size=(h, w))
            rgbe[0, 0, 3] = 0                                   # e = 0: black whatever the mantissas
            rgbe[1, 2:9 if w > 9 else 3] = rgbe[1, 2]           # a run
            with gltf.load_glb(_env_scene(tmp_path, gltf.encode_hdr(rgbe, coded))) as g:
                sc = g.scene_data()
            texels, intensity = sc.env_map
            assert sc.env is None and intensity == 1.5 and texels.shape == (h, w, 4)
            f = np.ldexp(np.float32(1.0), rgbe[..., 3].astype(np.int32) - 136).astype(np.float32)
            want = rgbe[..., :3].astype(np.float32) * f[..., None]
            want[rgbe[..., 3] == 0] = 0
            assert np.array_equal(texels[..., :3], want) and np.all(texels[..., 3] == 1.0)
        # failures: io/HDR.cpp:10-13 "Failed to load HDR file"
        good = gltf.encode_hdr(rng.randint(1, 256, size=(4, 16, 4)).astype(np.uint8))
        for bad in (good[:60], good.replace(b"32-bit_rle_rgbe", b"32-bit_rle_xyze"), good.replace(b"-Y 4 +X 16", b"+Y 4 +X 16"), b"P6\n"):
            with pytest.raises(api.RtgpuError, match="Failed to load HDR file"):
                gltf.load_glb(_env_scene(tmp_path, bad))
        with pytest.raises(api.RtgpuError, match="Failed to load HDR file"):
            gltf.load_glb(_env_scene(tmp_path, good, hdr_name="#WORKDIR#/missing.hdr"))
    finally:
        gltf.set_workdir("")


def _render_scene(tmp_path):
    """A small room exported to .glb: instanced spheres under a scaled parent, textured floor, lights from extras."""
    b = gltf.GlbBuilder()
    rng = np.random.RandomState(1)
    tex = b.add_texture(rng.randint(60, 256, size=(8, 8, 4)).astype(np.uint8))
    floor_m = b.add_material(pbrMetallicRoughness={"baseColorFactor": [1, 1, 1, 1], "baseColorTexture": {"index": tex}})
    ball_m = b.add_material(pbrMetallicRoughness={"baseColorFactor": [0.8, 0.3, 0.2, 1], "metallicFactor": 0.5, "roughnessFactor": 0.2})
    glow_m = b.add_material(pbrMetallicRoughness={"baseColorFactor": [0.2, 0.6, 0.3, 1]}, emissiveFactor=[0.1, 0.3, 0.6])
    fv, ff = scenes.grid_quad((-3, 0, -3), (0, 0, 6), (6, 0, 0), 3, 3)
    fv = np.asarray(fv, np.float32); ff = np.asarray(ff, np.uint32)
    floor = b.add_primitive_mesh([(fv, scenes.vertex_normals(fv, ff), ff, fv[:, [0, 2]] * 0.4, floor_m)])
    us = scenes.uv_sphere(12, 6)
    ball = b.add_primitive_mesh([(us.vertices, us.normals, us.indices, None, ball_m)])
    glow = b.add_primitive_mesh([(us.vertices, us.normals, us.indices, None, glow_m)])
    b.add_node(mesh=floor)
    group = b.add_node(translation=[0.0, 0.7, 0.0], scale=[0.7, 0.7, 0.7])
    b.add_node(parent=group, mesh=ball, translation=[-1.5, 0.0, 0.0])
    b.add_node(parent=group, mesh=ball, translation=[1.5, 0.0, 0.5], rotation=[0.0, 0.5, 0.0, 0.8660254])
    b.add_node(parent=group, mesh=glow, translation=[0.0, 0.6, -1.5], scale=[0.6, 0.6, 0.6], extras={"Material.EmissionStrength": 1.5})
    b.add_node(translation=[1.0, 4.0, 2.0], extras={"IsPointLight": True, "LightIntensity": 150.0})
    cam_m = scenes.look_at((0.0, 2.2, 5.0), (0.0, 0.6, 0.0))
    b.add_node(camera=b.add_camera(0.7), matrix=[float(x) for x in np.asarray(cam_m).T.reshape(-1)], extras={"IsMainCamera": True})
    b.set_scene_extras(EnvironmentIntensity=0.3, EnvironmentColor=[0.5, 0.6, 0.8])
    path = os.path.join(tmp_path, "room.glb")
    b.write(path)
    return path


def test_oracle_renders_a_loaded_file(tmp_path):
    """The flattened description is a complete scene: the oracle renders it, instanced and soupified alike."""
    path = _render_scene(str(tmp_path))
    st = abi.make_settings(32, 24, geometry_aa=4, material_aa=4, seed=3)
    with gltf.load_glb(path, 32 / 24) as g, gltf.load_glb(path, 32 / 24, soupify=True) as s:
        a, _ = oracle.OracleScene(g.scene_data()).render(g.camera(), st)
        b, _ = oracle.OracleScene(s.scene_data()).render(s.camera(), st)
    assert np.isfinite(a).all() and a.mean() > 0.05
    # same surfaces, same random numbers; normals of the soup are not renormalised after the scale (TriangleMesh.cpp:311-325)
    assert np.sqrt(np.mean((a - b) ** 2)) / np.sqrt(np.mean(a ** 2)) < 0.05


@pytest.mark.gpu
def test_gpu_renders_a_loaded_file_like_the_oracle(tmp_path):
    """File -> rtgpu_gltf_load -> rtgpu_scene_create(desc) -> render, against the oracle on the same description."""
    path = _render_scene(str(tmp_path))
    st = abi.make_settings(64, 48, geometry_aa=4, material_aa=8, seed=3)
    for soup in (False, True):
        with gltf.load_glb(path, 64 / 48, soupify=soup) as g:
            cam = g.camera()
            ref, _ = oracle.OracleScene(g.scene_data()).render(cam, st)
            with api.Scene(g.desc) as scene:
                img, _ = api.Renderer().render(scene, cam, st)
        assert np.isfinite(img).all()
        assert np.sqrt(np.mean((img - ref) ** 2)) / np.sqrt(np.mean(ref ** 2)) <= 1e-4


@pytest.mark.gpu
def test_render_glb_program_follows_the_reference_command_line(tmp_path):
    """examples/render_glb (the render branch of raytracer/src/exe/Raytracer.cpp with the same option names):
    a tile of the frame equals the same tile rendered through the Python mirror bit for bit; a file without a
    camera ends with the reference's message and exit code -1 (exe/Raytracer.cpp:224-226 style)."""
    import subprocess
    from raytracer_amd import tilefile
    exe = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "examples", "render_glb")
    if not os.path.exists(exe):
        pytest.fail("examples/render_glb is missing: build it with `make -C examples`")
    path = _render_scene(str(tmp_path))
    out = os.path.join(str(tmp_path), "frame")
    run = subprocess.run([exe, "--scene", path, "--width", "64", "--height", "48", "--xstart", "8", "--ystart", "4", "--xend", "56", "--yend", "40",
                          "--aageometry", "4", "--aamaterial", "4", "--seed", "9", "--outputtype", "tile", "--output", out],
                         capture_output=True, text=True, timeout=120)
    assert run.returncode == 0, run.stderr
    assert "Loading scene." in run.stdout and "rays: primary 6912" in run.stdout   # 48 x 36 pixels x 4
    w, h, tile, px = tilefile.read_tile_file(out + ".tile")
    assert (w, h, tile) == (64, 48, (8, 4, 56, 40))
    st = abi.make_settings(64, 48, geometry_aa=4, material_aa=4, seed=9, tile=(8, 4, 56, 40))
    with gltf.load_glb(path, 64 / 48) as g:
        with api.Scene(g.desc) as scene:
            img, _ = api.Renderer().render(scene, g.camera(), st)
    assert np.array_equal(px, img[4:40, 8:56]) and px.mean() > 0.05
    # --pmmode full: the photon map is built from the same arguments on both sides
    run = subprocess.run([exe, "--scene", path, "--width", "64", "--height", "48", "--aageometry", "2", "--aamaterial", "2", "--seed", "9",
                          "--pmmode", "full", "--pmdepth", "1", "--pmrayspointlamp", "20000", "--pmraysarealamp", "20000",
                          "--outputtype", "tile", "--output", out + "_pm"], capture_output=True, text=True, timeout=120)
    assert run.returncode == 0, run.stderr
    _, _, _, px_pm = tilefile.read_tile_file(out + "_pm.tile")
    with gltf.load_glb(path, 64 / 48) as g:
        with api.Scene(g.desc) as scene:
            n_photons = scene.build_photon_map(abi.PM_FULL, 1, 20000, 20000, seed=9)
            img_pm, _ = api.Renderer().render(scene, g.camera(), abi.make_settings(64, 48, geometry_aa=2, material_aa=2, seed=9))
    assert "Building photon KD-tree (%d photons)" % n_photons in run.stdout and n_photons > 1000   # an open scene: most rays leave it
    assert np.array_equal(px_pm, img_pm)
    # two tiles rendered by separate runs and merged (the reference's way of spreading a frame over processes)
    common = ["--scene", path, "--width", "64", "--height", "48", "--aageometry", "4", "--aamaterial", "4", "--seed", "9", "--outputtype", "tile"]
    for name, (xs, xe) in (("left", (0, 30)), ("right", (30, 64))):
        run = subprocess.run([exe] + common + ["--xstart", str(xs), "--xend", str(xe), "--output", out + "_" + name], capture_output=True, text=True, timeout=120)
        assert run.returncode == 0, run.stderr
    run = subprocess.run([exe, "--width", "64", "--height", "48", "--mergetiles", out + "_left.tile", out + "_right.tile", "--outputtype", "tile",
                          "--output", out + "_merged"], capture_output=True, text=True, timeout=120)
    assert run.returncode == 0 and "Merging tile files..." in run.stdout, run.stderr
    _, _, whole, px_merged = tilefile.read_tile_file(out + "_merged.tile")
    with gltf.load_glb(path, 64 / 48) as g:
        with api.Scene(g.desc) as scene:
            full, _ = api.Renderer().render(scene, g.camera(), abi.make_settings(64, 48, geometry_aa=4, material_aa=4, seed=9))
    assert whole == (0, 0, 64, 48) and np.array_equal(px_merged, full)
    run = subprocess.run([exe, "--width", "32", "--height", "48", "--mergetiles", out + "_left.tile"], capture_output=True, text=True, timeout=120)
    assert run.returncode != 0 and "Resolution mismatch" in run.stderr          # io/TileFile.cpp:46-49
    b = gltf.GlbBuilder()
    v, n, idx, uv = _quad()
    b.add_node(mesh=b.add_primitive_mesh([(v, n, idx, None, None)]))
    b.write(os.path.join(str(tmp_path), "nocam.glb"))
    run = subprocess.run([exe, "--scene", os.path.join(str(tmp_path), "nocam.glb"), "--width", "8", "--height", "8"], capture_output=True, text=True, timeout=120)
    assert run.returncode != 0 and "No camera in scene." in run.stderr


@pytest.mark.gpu
def test_render_glb_program_paths_and_photon_map_files(tmp_path):
    """The command line of the reference's render_sequence.py: --workdir with "#WORKDIR#" in the scene path
    (utility/PathResolver.cpp), an output path that already has an extension (kept, exe/Raytracer.cpp:152-155), output
    directories created (:345), --savepm / --loadpm / --pmfile (:193-206, 296-322): a render from a saved map equals the
    render that built it."""
    import subprocess
    from raytracer_amd import tilefile
    exe = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "examples", "render_glb")
    if not os.path.exists(exe):
        pytest.fail("examples/render_glb is missing: build it with `make -C examples`")
    work = os.path.join(str(tmp_path), "frames")
    os.makedirs(work)
    os.rename(_render_scene(str(tmp_path)), os.path.join(work, "165.glb"))
    pmfile = os.path.join(str(tmp_path), "photonmap")
    common = [exe, "--workdir", work, "--scene", "#WORKDIR#/165.glb", "--width", "48", "--height", "36", "--aageometry", "2", "--aamaterial", "2",
              "--seed", "3", "--outputtype", "tile", "--pmmode", "full", "--pmfile", pmfile]
    run = subprocess.run(common + ["--output", "out/built.tile", "--pmrayspointlamp", "20000", "--pmraysarealamp", "2E4", "--savepm"],
                         capture_output=True, text=True, timeout=120)
    assert run.returncode == 0, run.stderr
    assert "Building photon map..." in run.stdout and "Saving photon map..." in run.stdout and "Rendering..." in run.stdout
    built = os.path.join(work, "out", "built.tile")            # relative to the work directory, directory created, extension kept
    assert "Writing output to " + built in run.stdout and os.path.isfile(built)
    run = subprocess.run(common + ["--output", os.path.join(str(tmp_path), "loaded"), "--loadpm"], capture_output=True, text=True, timeout=120)
    assert run.returncode == 0, run.stderr
    n_photons = (os.path.getsize(pmfile) - 16) // 40
    assert "Loaded %d photons" % n_photons in run.stdout and n_photons > 1000
    a, b = tilefile.read_tile_file(built)[3], tilefile.read_tile_file(os.path.join(str(tmp_path), "loaded.tile"))[3]
    assert np.array_equal(a, b) and a.mean() > 0.01
    # the reference's error exits
    for args, message in ((["--workdir", os.path.join(work, "nowhere")], "The specified work directory does not exist or is not a folder."),
                          (["--workdir", work, "--scene", "166.glb"], "The specified scene file does not exist or is not a file."),
                          (["--workdir", work, "--scene", "165.glb", "--loadpm", "--pmfile", pmfile + "x"], "The specified photon map file does not exist or is not a file."),
                          (["--workdir", work, "--scene", "165.glb", "--pmmode", "some"], "Invalid photonmapping mode!"),
                          (["--workdir", work, "--scene", "165.glb", "--aamaterial", "0"], "Invalid AA settings!"),
                          (["--workdir", work, "--scene", "165.glb", "--pmraysarealamp", "0"], "Invalid PM raycount settings!"),
                          (["--workdir", work, "--scene", "165.glb", "--samples", "4"], "unrecognised option")):
        run = subprocess.run([exe] + args, capture_output=True, text=True, timeout=120)
        assert run.returncode != 0 and message in run.stderr, (args, run.stderr)
    open(pmfile + "_bad", "wb").write(b"RTGPUPM1" + struct.pack("<Q", 1 << 40))
    run = subprocess.run(common[:-1] + [pmfile + "_bad", "--loadpm"], capture_output=True, text=True, timeout=120)
    assert run.returncode != 0 and "truncated" in run.stderr


def test_damaged_files_are_refused_not_crashed_on(tmp_path):
    """Bytes of a valid file overwritten at random (JSON chunk, chunk headers, binary chunk): the loader either
    still reads a scene or returns an error code; it never reads outside its buffers."""
    path, _, _ = _full_scene(str(tmp_path))
    data = bytearray(open(path, "rb").read())
    json_len = struct.unpack_from("<I", data, 12)[0]
    rng = np.random.RandomState(7)
    bad = os.path.join(str(tmp_path), "damaged.glb")
    outcomes = {"ok": 0, "error": 0}
    for k in range(400):
        d = bytearray(data)
        region = (12, 20 + json_len) if k % 2 == 0 else (0, len(d))
        for _ in range(rng.randint(1, 6)):
            pos = rng.randint(region[0], region[1])
            d[pos] = rng.randint(0, 256) if k % 3 else ord(rng.choice(list('{}[]",:0129-e.')))
        if k % 7 == 0:
            d = d[:rng.randint(12, len(d))]
        open(bad, "wb").write(bytes(d))
        try:
            with gltf.load_glb(bad, 1.5, soupify=bool(k % 2)) as g:
                g.scene_data()
                if g.camera_count():
                    g.camera()
            outcomes["ok"] += 1
        except (api.RtgpuError, ValueError):
            outcomes["error"] += 1
    assert outcomes["error"] > 50 and outcomes["ok"] + outcomes["error"] == 400


def _rewrite_json(path, out_path, edit):
    """A copy of a .glb whose JSON chunk went through `edit` (the binary chunk is kept)."""
    import json
    data = open(path, "rb").read()
    json_len = struct.unpack_from("<I", data, 12)[0]
    doc = json.loads(data[20:20 + json_len].decode("utf-8"))
    edit(doc)
    js = json.dumps(doc).encode("utf-8")
    js += b" " * (-len(js) % 4)
    rest = data[20 + json_len:]
    body = struct.pack("<II", len(js), 0x4E4F534A) + js + rest
    open(out_path, "wb").write(struct.pack("<III", 0x46546C67, 2, 12 + len(body)) + body)


@pytest.mark.parametrize("field,value", [("view.byteLength", -1), ("view.byteOffset", -8), ("accessor.byteOffset", -12),
                                         ("view.byteLength", 1e300), ("view.byteOffset", 2.0 ** 63), ("view.byteStride", -4),
                                         ("accessor.byteOffset", 0.5), ("view.byteOffset", float("nan"))])
def test_negative_huge_and_fractional_byte_counts_are_refused(tmp_path, field, value):
    """Offsets, lengths and strides are JSON numbers: negative, non-finite, fractional or huge ones must be an error,
    not a wrapped sum that passes the bounds check (a byteLength of -1 with an accessor offset of -12 used to read in
    front of the buffer)."""
    path, _, _ = _full_scene(str(tmp_path))
    bad = os.path.join(str(tmp_path), "bad_counts.glb")

    def edit(doc):
        where, name = field.split(".")
        for obj in (doc["bufferViews"] if where == "view" else doc["accessors"]):
            obj[name] = value
    _rewrite_json(path, bad, edit)   # (Python writes a NaN as the bare word NaN, as sloppy exporters do)
    with pytest.raises((api.RtgpuError, ValueError)):
        with gltf.load_glb(bad, 1.5, soupify=False) as g:
            g.scene_data()


def test_loader_under_address_and_undefined_behaviour_sanitizers(tmp_path):
    """tests/native/fuzz_gltf.cpp: the loader's source built for the host with -fsanitize=address,undefined and run on
    1500 damaged copies of a valid file; any out-of-bounds access or undefined operation aborts the harness."""
    import shutil
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if not shutil.which("g++") or not os.path.exists("/opt/rocm/include/hip/hip_runtime.h"):
        pytest.skip("needs g++ and the HIP headers")
    exe = os.path.join(str(tmp_path), "fuzz_gltf")
    build = subprocess.run(["g++", "-std=c++17", "-g", "-O1", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined", "-D__HIP_PLATFORM_AMD__",
                            "-I/opt/rocm/include", "-I" + os.path.join(root, "raytracer_amd", "csrc"), os.path.join(root, "tests", "native", "fuzz_gltf.cpp"),
                            "-o", exe, "-lz"], capture_output=True, text=True, timeout=600)
    assert build.returncode == 0, build.stderr[-2000:]
    path, _, _ = _full_scene(str(tmp_path))
    run = subprocess.run([exe, path, os.path.join(str(tmp_path), "scratch.glb"), "1500"], capture_output=True, text=True, timeout=600)
    assert run.returncode == 0, (run.stdout + run.stderr)[-3000:]
    assert "ok " in run.stdout and "err " in run.stdout
    # the Radiance reader: a scene that names scratch.hdr, damaged copies of a run-length coded file written there
    rng = np.random.RandomState(9)
    rgbe = rng.randint(0, 256, size=(12, 24, 4)).astype(np.uint8)
    rgbe[3:6, 4:20] = rgbe[3, 4]
    good = os.path.join(str(tmp_path), "good.hdr")
    open(good, "wb").write(gltf.encode_hdr(rgbe))
    scene = _env_scene(tmp_path, b"", hdr_name=os.path.join(str(tmp_path), "scratch.hdr"))
    run = subprocess.run([exe, good, os.path.join(str(tmp_path), "scratch.hdr"), "1500", scene], capture_output=True, text=True, timeout=600)
    assert run.returncode == 0, (run.stdout + run.stderr)[-3000:]
    ok, err = [int(x) for x in run.stdout.split()[1::2]]
    assert ok > 100 and err > 100


# ---- files not written by this repository's GlbBuilder ------------------------------------------------------
# The JSON and the base64 buffers below are the published Khronos glTF tutorial samples "a minimal glTF file"
# (gltfTutorial_003: one indexed triangle, positions only) and "simple meshes" (gltfTutorial_008: the same triangle
# with normals, used by two nodes, the second translated by (1, 0, 0)), typed in from the tutorial; the .glb container
# is assembled here byte by byte from the glTF 2.0 specification (12-byte header, JSON chunk padded with spaces, BIN
# chunk padded with zeros), not by GlbBuilder.
_KHRONOS_MINIMAL = """{
  "scene": 0, "scenes": [{"nodes": [0]}], "nodes": [{"mesh": 0}],
  "meshes": [{"primitives": [{"attributes": {"POSITION": 1}, "indices": 0}]}],
  "buffers": [{"byteLength": 44}],
  "bufferViews": [{"buffer": 0, "byteOffset": 0, "byteLength": 6, "target": 34963},
                  {"buffer": 0, "byteOffset": 8, "byteLength": 36, "target": 34962}],
  "accessors": [{"bufferView": 0, "byteOffset": 0, "componentType": 5123, "count": 3, "type": "SCALAR", "max": [2], "min": [0]},
                {"bufferView": 1, "byteOffset": 0, "componentType": 5126, "count": 3, "type": "VEC3", "max": [1.0, 1.0, 0.0], "min": [0.0, 0.0, 0.0]}],
  "asset": {"version": "2.0"}
}"""
_KHRONOS_MINIMAL_BIN = "AAABAAIAAAAAAAAAAAAAAAAAAAAAAIA/AAAAAAAAAAAAAAAAAACAPwAAAAA="
_KHRONOS_SIMPLE_MESHES = """{
  "scene": 0, "scenes": [{"nodes": [0, 1]}],
  "nodes": [{"mesh": 0}, {"mesh": 0, "translation": [1.0, 0.0, 0.0]}],
  "meshes": [{"primitives": [{"attributes": {"POSITION": 1, "NORMAL": 2}, "indices": 0}]}],
  "buffers": [{"byteLength": 80}],
  "bufferViews": [{"buffer": 0, "byteOffset": 0, "byteLength": 6, "target": 34963},
                  {"buffer": 0, "byteOffset": 8, "byteLength": 72, "byteStride": 12, "target": 34962}],
  "accessors": [{"bufferView": 0, "byteOffset": 0, "componentType": 5123, "count": 3, "type": "SCALAR", "max": [2], "min": [0]},
                {"bufferView": 1, "byteOffset": 0, "componentType": 5126, "count": 3, "type": "VEC3", "max": [1.0, 1.0, 0.0], "min": [0.0, 0.0, 0.0]},
                {"bufferView": 1, "byteOffset": 36, "componentType": 5126, "count": 3, "type": "VEC3", "max": [0.0, 0.0, 1.0], "min": [0.0, 0.0, 1.0]}],
  "asset": {"version": "2.0"}
}"""
_KHRONOS_SIMPLE_MESHES_BIN = ("AAABAAIAAAAAAAAAAAAAAAAAAAAAAIA/AAAAAAAAAAAAAAAAAACAPwAAAAAAAAAAAAAAAAAAgD8AAAAAAAAAAAAAgD8AAAAAAAAAAAAAgD8=")


def _glb_by_hand(json_text, bin_b64, path):
    import base64
    js = json_text.encode("utf-8")
    js += b" " * (-len(js) % 4)
    blob = base64.b64decode(bin_b64)
    blob += b"\0" * (-len(blob) % 4)
    total = 12 + 8 + len(js) + 8 + len(blob)
    with open(path, "wb") as f:
        f.write(struct.pack("<4sII", b"glTF", 2, total))
        f.write(struct.pack("<I4s", len(js), b"JSON") + js)
        f.write(struct.pack("<I4s", len(blob), b"BIN\0") + blob)
    return path


def test_khronos_tutorial_files_assembled_by_hand(tmp_path):
    """Independent inputs for the bridge.  The minimal triangle has no normals: loadGLTFScene refuses it with
    "Implicit normals are not supported" (io/GLTF.cpp:229-238).  The simple-meshes file -- whose accessors share a buffer
    view, which files written by GlbBuilder never do -- loads: one mesh stored once
    (meshCache, :203-216), two instances (one per node and primitive, DynamicScene.cpp:99-161), the second moved by
    (1, 0, 0); no camera ("No camera in scene.", renderer/Renderer.cpp:318), default material (white diffuse)."""
    p = str(tmp_path)
    with pytest.raises(api.RtgpuError, match="Implicit normals are not supported"):
        gltf.load_glb(_glb_by_hand(_KHRONOS_MINIMAL, _KHRONOS_MINIMAL_BIN, os.path.join(p, "minimal.glb")))
    with gltf.load_glb(_glb_by_hand(_KHRONOS_SIMPLE_MESHES, _KHRONOS_SIMPLE_MESHES_BIN, os.path.join(p, "simple.glb"))) as g:
        sc = g.scene_data()
        assert g.camera_count() == 0
        with pytest.raises(api.RtgpuError, match="No camera in scene."):
            g.camera()
    assert len(sc.meshes) == 1 and len(sc.instances) == 2 and [i.shape for i in sc.instances] == [0, 0]
    m = sc.meshes[0]
    # positions and normals share one buffer view: loadVec3s walks from the accessor's start to the end of the *view*
    # whatever the accessor's count says (io/GLTF.cpp:94-109), so the reference reads six "positions" (the three normals
    # after the three positions) and three normals; the triangle only names vertices 0..2 of either array
    assert np.array_equal(m.vertices, np.array([[0, 0, 0], [1, 0, 0], [0, 1, 0], [0, 0, 1], [0, 0, 1], [0, 0, 1]], np.float32))
    assert np.array_equal(m.normals[:3], np.array([[0, 0, 1]] * 3, np.float32)) and len(m.normals) == 6
    assert np.array_equal(np.asarray(m.indices).reshape(-1), [0, 1, 2])
    assert np.array_equal(sc.instances[0].transform, np.eye(4)) and np.array_equal(sc.instances[1].transform, scenes.translate(1.0, 0.0, 0.0))
    assert np.array_equal(sc.instances[1].inverse, scenes.translate(-1.0, 0.0, 0.0))
    assert len(sc.point_lights) == 0 and len(sc.area_lights) == 0
    # the oracle traces the loaded scene: a ray down -z above each triangle's inside hits instance 0 / 1 at t = 1
    S = oracle.OracleScene(sc)
    o = np.array([[0.25, 0.25, 1.0], [1.25, 0.25, 1.0], [0.9, 0.9, 1.0]], np.float32)
    d = np.tile(np.array([[0.0, 0.0, -1.0]], np.float32), (3, 1))
    h = S.trace_closest(o, d)
    assert list(h["inst"]) == [0, 1, abi.INVALID_ID] and list(h["t"][:2]) == [1.0, 1.0] and list(h["prim"][:2]) == [0, 0]
