"""Binary glTF in and out.

`load_glb` mirrors loadGLTFScene + DynamicScene::build / soupifyScene (raytracer/src/lib/io/GLTF.cpp:682-705,
scene/dynamic/DynamicScene.cpp:17-160) through the C ABI's `rtgpu_gltf_*` entry points: the file is parsed
and flattened by the library (host code), and the result is the `rtgpu_scene_desc` that `rtgpu_scene_create`
takes.  `GltfScene.scene_data()` copies that description into numpy (`abi.SceneData`) so the same scene can be
given to `api.Scene` and, in the tests, to the CPU oracle.

`write_glb` writes the subset of glTF the reference's loader reads -- meshes with NORMAL / TEXCOORD_0, node
hierarchies with TRS or matrices, pbrMetallicRoughness materials, PNG textures and the Blender "extras" the
loader interprets (lights, camera and material parameters) -- for tests and for exporting procedural scenes.
"""
from __future__ import annotations

import ctypes as C
import json
import struct
import zlib

import numpy as np

from . import abi, api


def _lib():
    L = api.lib()
    if not getattr(L, "_gltf_ready", False):
        L.rtgpu_gltf_load.argtypes = [C.c_char_p, C.c_float, C.c_int, C.POINTER(C.c_void_p)]
        L.rtgpu_gltf_scene_desc.argtypes = [C.c_void_p]
        L.rtgpu_gltf_scene_desc.restype = C.POINTER(abi.SceneDesc)
        L.rtgpu_gltf_camera.argtypes = [C.c_void_p, C.POINTER(abi.Camera)]
        L.rtgpu_gltf_camera_count.argtypes = [C.c_void_p]
        L.rtgpu_gltf_camera_count.restype = C.c_uint32
        L.rtgpu_gltf_free.argtypes = [C.c_void_p]
        L.rtgpu_gltf_free.restype = None
        L.rtgpu_gltf_set_workdir.argtypes = [C.c_char_p]
        L.rtgpu_gltf_set_workdir.restype = None
        L._gltf_ready = True
    return L


class GltfScene:
    """A loaded .glb file: `desc` is the rtgpu_scene_desc (valid until close())."""

    def __init__(self, path, image_aspect_ratio=1.0, soupify=False):
        h = C.c_void_p()
        api._check(_lib().rtgpu_gltf_load(str(path).encode(), float(image_aspect_ratio), 1 if soupify else 0, C.byref(h)))
        self._h = h

    def close(self):
        if getattr(self, "_h", None):
            _lib().rtgpu_gltf_free(self._h)
            self._h = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def desc(self) -> abi.SceneDesc:
        return _lib().rtgpu_gltf_scene_desc(self._h).contents

    def camera_count(self) -> int:
        return int(_lib().rtgpu_gltf_camera_count(self._h))

    def camera(self) -> abi.Camera:
        """Renderer::findCamera (renderer/Renderer.cpp:304-320); RuntimeError("No camera in scene.") without one."""
        cam = abi.Camera()
        api._check(_lib().rtgpu_gltf_camera(self._h, C.byref(cam)))
        return cam

    def scene_data(self) -> abi.SceneData:
        d = self.desc
        sc = abi.SceneData()
        for i in range(d.n_meshes):
            m = d.meshes[i]
            nv, nt = int(m.n_vertices), int(m.n_triangles)
            arr = lambda p, n, dt: np.ctypeslib.as_array(p, shape=(n,)).astype(dt, copy=True) if n else np.zeros(0, dt)
            sc.meshes.append(abi.MeshData(arr(m.vertices, nv * 3, np.float32).reshape(-1, 3), arr(m.normals, nv * 3, np.float32).reshape(-1, 3),
                                          arr(m.indices, nt * 3, np.uint32).reshape(-1, 3),
                                          arr(m.tri_material, nt, np.uint32) if m.tri_material else None,
                                          arr(m.texcoords, nv * 2, np.float32).reshape(-1, 2) if m.texcoords else None))
        for i in range(d.n_instances):
            it = d.instances[i]
            sc.instances.append(abi.InstanceData(int(it.shape), int(it.material), np.array(it.transform[:]).reshape(4, 4),
                                                 np.array(it.inverse[:]).reshape(4, 4)))
        for i in range(d.n_materials):
            m = d.materials[i]
            sc.materials.append(abi.MaterialData(int(m.type), tuple(m.color[:]), float(m.intensity), float(m.roughness), float(m.mix_factor),
                                                 int(m.first), int(m.second), int(m.texture) - 1 if m.texture else None,
                                                 int(m.normal_map) - 1 if m.normal_map else None))
        for i in range(d.n_point_lights):
            l = d.point_lights[i]
            sc.point_lights.append((tuple(l.pos[:]), tuple(l.color[:]), float(l.intensity)))
        for i in range(d.n_area_lights):
            l = d.area_lights[i]
            sc.area_lights.append((tuple(l.a[:]), tuple(l.b[:]), tuple(l.c[:]), tuple(l.color[:]), float(l.intensity)))
        for i in range(d.n_directional_lights):
            l = d.directional_lights[i]
            sc.directional_lights.append((tuple(l.direction[:]), float(l.angle), tuple(l.color[:]), float(l.intensity)))
        for i in range(d.n_textures):
            t = d.textures[i]
            px = np.ctypeslib.as_array(t.rgba, shape=(int(t.height), int(t.width), 4)).copy()
            sc.textures.append((px, float(t.gamma)))
        if d.has_environment == 1:
            sc.env = (tuple(d.env_color[:]), float(d.env_intensity))
        elif d.has_environment == 2:
            n = int(d.env_map_height) * int(d.env_map_width) * 4
            texels = np.ctypeslib.as_array(d.env_map, shape=(n,)).reshape(int(d.env_map_height), int(d.env_map_width), 4).copy()
            sc.env_map = (texels, float(d.env_intensity))
        return sc


def set_workdir(directory):
    """PathResolver::setWorkDir: what "#WORKDIR#" stands for in the paths a file names (process-wide)."""
    _lib().rtgpu_gltf_set_workdir(str(directory).encode())


def encode_hdr(pixels_rgbe, coded=True) -> bytes:
    """A Radiance .hdr file of (height, width, 4) RGBE bytes: per-component run-length coded rows when `coded` and
    8 <= width < 32768, flat otherwise."""
    px = np.ascontiguousarray(pixels_rgbe, dtype=np.uint8)
    h, w = px.shape[:2]
    out = bytearray(b"#?RADIANCE\nFORMAT=32-bit_rle_rgbe\n\n" + ("-Y %d +X %d\n" % (h, w)).encode())
    if not (coded and 8 <= w < 32768):
        return bytes(out) + px.tobytes()
    for y in range(h):
        out += bytes([2, 2, w >> 8, w & 255])
        for k in range(4):
            row = px[y, :, k]
            i = 0
            while i < w:
                run = 1
                while i + run < w and run < 127 and row[i + run] == row[i]:
                    run += 1
                if run >= 3:
                    out += bytes([128 + run, int(row[i])])
                    i += run
                else:
                    j = i
                    while j < w and j - i < 128 and not (j + 2 < w and row[j] == row[j + 1] == row[j + 2]):
                        j += 1
                    j = max(j, i + 1)
                    out += bytes([j - i]) + row[i:j].tobytes()
                    i = j
    return bytes(out)


def load_glb(path, image_aspect_ratio=1.0, soupify=False) -> GltfScene:
    return GltfScene(path, image_aspect_ratio, soupify)


# ---------------------------------------------------------------------------------------------
# writer
# ---------------------------------------------------------------------------------------------
def encode_png(pixels) -> bytes:
    """8-bit RGBA PNG (filter 0 rows, one IDAT)."""
    px = np.ascontiguousarray(pixels, dtype=np.uint8)
    if px.ndim != 3 or px.shape[2] != 4:
        raise ValueError("pixels must have shape (height, width, 4)")
    h, w = px.shape[:2]
    raw = b"".join(b"\x00" + px[y].tobytes() for y in range(h))

    def chunk(tag, body):
        return struct.pack(">I", len(body)) + tag + body + struct.pack(">I", zlib.crc32(tag + body) & 0xFFFFFFFF)

    return (b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, 6, 0, 0, 0)) + chunk(b"IDAT", zlib.compress(raw, 6))
            + chunk(b"IEND", b""))


class GlbBuilder:
    """Collects buffers, accessors, meshes, materials and nodes and writes one .glb file."""

    def __init__(self):
        self.bin = bytearray()
        self.doc = {"asset": {"version": "2.0"}, "scene": 0, "scenes": [{"nodes": []}], "nodes": [], "meshes": [], "materials": [],
                    "accessors": [], "bufferViews": [], "cameras": [], "images": [], "textures": []}

    def _view(self, data: bytes) -> int:
        while len(self.bin) % 4:
            self.bin.append(0)
        off = len(self.bin)
        self.bin += data
        self.doc["bufferViews"].append({"buffer": 0, "byteOffset": off, "byteLength": len(data)})
        return len(self.doc["bufferViews"]) - 1

    def _accessor(self, array, component_type, type_name) -> int:
        a = np.ascontiguousarray(array)
        self.doc["accessors"].append({"bufferView": self._view(a.tobytes()), "componentType": component_type, "count": int(a.shape[0]),
                                      "type": type_name})
        return len(self.doc["accessors"]) - 1

    def add_primitive_mesh(self, primitives) -> int:
        """primitives: list of (vertices, normals, indices, texcoords or None, material index or None)."""
        prims = []
        for v, n, idx, uv, mat in primitives:
            idx = np.asarray(idx).reshape(-1)
            it = (np.uint16, 5123) if (idx.size == 0 or int(idx.max()) < 65536) else (np.uint32, 5125)
            attrs = {"POSITION": self._accessor(np.asarray(v, np.float32).reshape(-1, 3), 5126, "VEC3"),
                     "NORMAL": self._accessor(np.asarray(n, np.float32).reshape(-1, 3), 5126, "VEC3")}
            if uv is not None:
                attrs["TEXCOORD_0"] = self._accessor(np.asarray(uv, np.float32).reshape(-1, 2), 5126, "VEC2")
            p = {"attributes": attrs, "indices": self._accessor(idx.astype(it[0]), it[1], "SCALAR"), "mode": 4}
            if mat is not None:
                p["material"] = int(mat)
            prims.append(p)
        self.doc["meshes"].append({"primitives": prims})
        return len(self.doc["meshes"]) - 1

    def add_texture(self, pixels) -> int:
        self.doc["images"].append({"bufferView": self._view(encode_png(pixels)), "mimeType": "image/png"})
        self.doc["textures"].append({"source": len(self.doc["images"]) - 1})
        return len(self.doc["textures"]) - 1

    def add_material(self, **fields) -> int:
        """fields: glTF material members (pbrMetallicRoughness, normalTexture, emissiveFactor, emissiveTexture, ...)."""
        self.doc["materials"].append(dict(fields))
        return len(self.doc["materials"]) - 1

    def add_camera(self, yfov) -> int:
        self.doc["cameras"].append({"type": "perspective", "perspective": {"yfov": float(yfov), "znear": 0.01}})
        return len(self.doc["cameras"]) - 1

    def add_node(self, parent=None, **fields) -> int:
        """fields: mesh, camera, translation, rotation (x, y, z, w), scale, matrix (column-major 16), extras."""
        self.doc["nodes"].append(dict(fields))
        idx = len(self.doc["nodes"]) - 1
        if parent is None:
            self.doc["scenes"][0]["nodes"].append(idx)
        else:
            self.doc["nodes"][parent].setdefault("children", []).append(idx)
        return idx

    def set_scene_extras(self, **extras):
        self.doc["scenes"][0]["extras"] = dict(extras)

    def write(self, path):
        doc = {k: v for k, v in self.doc.items() if not (isinstance(v, list) and not v)}
        doc["buffers"] = [{"byteLength": len(self.bin)}]
        js = json.dumps(doc).encode()
        js += b" " * (-len(js) % 4)
        body = bytes(self.bin) + b"\x00" * (-len(self.bin) % 4)
        total = 12 + 8 + len(js) + 8 + len(body)
        with open(path, "wb") as f:
            f.write(struct.pack("<III", 0x46546C67, 2, total))
            f.write(struct.pack("<II", len(js), 0x4E4F534A) + js)
            f.write(struct.pack("<II", len(body), 0x004E4942) + body)
