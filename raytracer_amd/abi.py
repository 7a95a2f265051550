"""ctypes mirror of include/rtgpu.h (struct layouts and the scene-description packer).

The structs here are byte-for-byte the C structs of the C ABI, so one packed scene
description can be handed to any library that speaks that ABI.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass, field
from typing import List, Optional, Sequence

import numpy as np

INVALID_ID = 0xFFFFFFFF
SHAPE_SPHERE = -1

MAT_DIFFUSE, MAT_GLOSSY, MAT_EMISSIVE, MAT_CONSTMIX, MAT_FRESNELMIX, MAT_TRANSPARENT, MAT_GLASS, MAT_ADD = 0, 1, 2, 3, 4, 5, 6, 7
JITTER_RANDOM, JITTER_CENTER = 0, 1

c_float_p = C.POINTER(C.c_float)
c_u32_p = C.POINTER(C.c_uint32)
c_u8_p = C.POINTER(C.c_uint8)


class MeshDesc(C.Structure):
    _fields_ = [("vertices", c_float_p), ("normals", c_float_p), ("indices", c_u32_p),
                ("tri_material", c_u32_p), ("n_vertices", C.c_uint32), ("n_triangles", C.c_uint32),
                ("texcoords", c_float_p)]


class InstanceDesc(C.Structure):
    _fields_ = [("shape", C.c_int32), ("material", C.c_uint32),
                ("transform", C.c_double * 16), ("inverse", C.c_double * 16)]


class Material(C.Structure):
    _fields_ = [("type", C.c_int32), ("color", C.c_float * 3), ("intensity", C.c_double),
                ("roughness", C.c_float), ("texture", C.c_uint32), ("mix_factor", C.c_double),
                ("first", C.c_uint32), ("second", C.c_uint32), ("normal_map", C.c_uint32), ("reserved", C.c_uint32)]


class PointLight(C.Structure):
    _fields_ = [("pos", C.c_float * 3), ("color", C.c_float * 3), ("intensity", C.c_double)]


class AreaLight(C.Structure):
    _fields_ = [("a", C.c_float * 3), ("b", C.c_float * 3), ("c", C.c_float * 3),
                ("color", C.c_float * 3), ("intensity", C.c_double)]


class DirectionalLight(C.Structure):
    _fields_ = [("direction", C.c_float * 3), ("angle", C.c_float), ("color", C.c_float * 3), ("reserved", C.c_float),
                ("intensity", C.c_double)]


class Texture(C.Structure):
    _fields_ = [("rgba", c_u8_p), ("width", C.c_uint32), ("height", C.c_uint32), ("gamma", C.c_double)]


class Photon(C.Structure):
    _fields_ = [("pos", C.c_float * 3), ("normal", C.c_float * 3), ("energy", C.c_float * 3), ("is_caustic", C.c_uint32)]


class PhotonGather(C.Structure):
    _fields_ = [("count", C.c_uint32), ("max_dist", C.c_float), ("energy", C.c_float * 3)]


PHOTON_DTYPE = np.dtype([("pos", np.float32, 3), ("normal", np.float32, 3), ("energy", np.float32, 3), ("is_caustic", np.uint32)])
GATHER_DTYPE = np.dtype([("count", np.uint32), ("max_dist", np.float32), ("energy", np.float32, 3)])
PM_NONE, PM_CAUSTICS, PM_FULL = 0, 1, 2


class SceneDesc(C.Structure):
    _fields_ = [("meshes", C.POINTER(MeshDesc)), ("instances", C.POINTER(InstanceDesc)),
                ("materials", C.POINTER(Material)), ("point_lights", C.POINTER(PointLight)),
                ("area_lights", C.POINTER(AreaLight)),
                ("n_meshes", C.c_uint32), ("n_instances", C.c_uint32), ("n_materials", C.c_uint32),
                ("n_point_lights", C.c_uint32), ("n_area_lights", C.c_uint32),
                ("has_environment", C.c_int32), ("env_color", C.c_float * 3), ("env_intensity", C.c_double),
                ("directional_lights", C.POINTER(DirectionalLight)), ("n_directional_lights", C.c_uint32),
                ("n_textures", C.c_uint32), ("textures", C.POINTER(Texture)),
                ("env_map", C.POINTER(C.c_float)), ("env_map_width", C.c_uint32), ("env_map_height", C.c_uint32)]


class Hit(C.Structure):
    _fields_ = [("t", C.c_float), ("prim", C.c_uint32), ("inst", C.c_uint32),
                ("beta", C.c_float), ("gamma", C.c_float)]


HIT_DTYPE = np.dtype([("t", np.float32), ("prim", np.uint32), ("inst", np.uint32),
                      ("beta", np.float32), ("gamma", np.float32)])
assert HIT_DTYPE.itemsize == C.sizeof(Hit) == 20


class Camera(C.Structure):
    _fields_ = [("origin", C.c_float * 3), ("u", C.c_float * 3), ("v", C.c_float * 3), ("w", C.c_float * 3),
                ("width", C.c_double), ("aperture", C.c_double), ("focal_distance", C.c_double)]


class RenderSettings(C.Structure):
    _fields_ = [("width", C.c_int32), ("height", C.c_int32),
                ("x_start", C.c_int32), ("y_start", C.c_int32), ("x_end", C.c_int32), ("y_end", C.c_int32),
                ("geometry_aa", C.c_int32), ("material_aa", C.c_int32), ("max_path_length", C.c_int32),
                ("seed", C.c_uint32), ("jitter_mode", C.c_int32), ("n_gpus", C.c_int32)]


class RenderWork(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in ("extension_rays", "extension_boxes", "extension_triangles",
                                          "shadow_rays", "shadow_boxes", "shadow_triangles",
                                          "primary_rays", "primary_boxes", "primary_triangles")]


class RenderProfile(C.Structure):
    _fields_ = [(n, C.c_double) for n in ("primary_ms", "shade_ms", "trace_ms", "gather_ms", "resolve_ms", "total_ms")] + \
               [(n, C.c_uint32) for n in ("primary_launches", "shade_launches", "trace_launches", "gather_launches", "resolve_launches", "reserved")]


class RenderStats(C.Structure):
    _fields_ = [("primary_rays", C.c_uint64), ("extension_rays", C.c_uint64), ("shadow_rays", C.c_uint64),
                ("device_ms", C.c_double)]


class SceneInfo(C.Structure):
    _fields_ = [("n_triangles", C.c_uint64), ("n_nodes", C.c_uint64), ("n_leaves", C.c_uint64),
                ("max_depth", C.c_uint32), ("max_leaf_size", C.c_uint32), ("device_bytes", C.c_uint64),
                ("build_ms", C.c_double)]


PROGRESS_CB = C.CFUNCTYPE(None, C.c_char_p, C.c_float, C.c_int, C.c_void_p)


# ---------------------------------------------------------------------------------------
# Python-side scene description (numpy arrays) and its packing into SceneDesc
# ---------------------------------------------------------------------------------------
@dataclass
class MeshData:
    vertices: np.ndarray                     # (nv, 3) float32
    normals: np.ndarray                      # (nv, 3) float32
    indices: np.ndarray                      # (nt, 3) uint32
    tri_material: Optional[np.ndarray] = None  # (nt,) uint32
    texcoords: Optional[np.ndarray] = None     # (nv, 2) float32


@dataclass
class InstanceData:
    shape: int                               # mesh index or SHAPE_SPHERE
    material: int
    transform: np.ndarray                    # (4, 4) float64, object -> world
    inverse: Optional[np.ndarray] = None     # (4, 4) float64 (computed when omitted)


@dataclass
class MaterialData:
    type: int
    color: Sequence[float] = (1.0, 1.0, 1.0)
    intensity: float = 1.0
    roughness: float = 0.0
    mix_factor: float = 0.0
    first: int = 0
    second: int = 0
    texture: Optional[int] = None   # index into SceneData.textures (albedo of a diffuse, emission of an emissive material)
    normal_map: Optional[int] = None   # index into SceneData.textures (diffuse / glossy)


@dataclass
class SceneData:
    meshes: List[MeshData] = field(default_factory=list)
    instances: List[InstanceData] = field(default_factory=list)
    materials: List[MaterialData] = field(default_factory=list)
    point_lights: List[tuple] = field(default_factory=list)   # (pos3, color3, intensity)
    area_lights: List[tuple] = field(default_factory=list)    # (a3, b3, c3, color3, intensity)
    directional_lights: List[tuple] = field(default_factory=list)  # (direction3, angle, color3, intensity)
    textures: List[tuple] = field(default_factory=list)       # ((h, w, 4) uint8 pixels, gamma)
    env: Optional[tuple] = None                               # (color3, intensity)
    env_map: Optional[tuple] = None                           # ((height, width, 4) float32 texels, intensity): an image-map environment
    sky: bool = False                                         # SkyEnvironment (no parameters)

    def n_triangles(self) -> int:
        return int(sum(m.indices.shape[0] for m in self.meshes))


class PackedScene:
    """Holds a SceneDesc plus every buffer it points into (keeps them alive)."""

    def __init__(self, scene: SceneData):
        self._keep = []
        n_meshes = len(scene.meshes)
        meshes = (MeshDesc * max(1, n_meshes))()
        for i, m in enumerate(scene.meshes):
            v = np.ascontiguousarray(m.vertices, dtype=np.float32).reshape(-1, 3)
            n = np.ascontiguousarray(m.normals, dtype=np.float32).reshape(-1, 3)
            idx = np.ascontiguousarray(m.indices, dtype=np.uint32).reshape(-1, 3)
            if n.shape != v.shape:
                raise ValueError("normals must be per vertex")
            if idx.size and int(idx.max()) >= v.shape[0]:
                raise ValueError("triangle index out of range")
            self._keep += [v, n, idx]
            meshes[i].vertices = v.ctypes.data_as(c_float_p)
            meshes[i].normals = n.ctypes.data_as(c_float_p)
            meshes[i].indices = idx.ctypes.data_as(c_u32_p)
            if m.tri_material is not None:
                tm = np.ascontiguousarray(m.tri_material, dtype=np.uint32).reshape(-1)
                if tm.shape[0] != idx.shape[0]:
                    raise ValueError("tri_material must have one entry per triangle")
                self._keep.append(tm)
                meshes[i].tri_material = tm.ctypes.data_as(c_u32_p)
            if m.texcoords is not None:
                uv = np.ascontiguousarray(m.texcoords, dtype=np.float32).reshape(-1, 2)
                if uv.shape[0] != v.shape[0]:
                    raise ValueError("texcoords must be per vertex")
                self._keep.append(uv)
                meshes[i].texcoords = uv.ctypes.data_as(c_float_p)
            meshes[i].n_vertices = v.shape[0]
            meshes[i].n_triangles = idx.shape[0]
        n_inst = len(scene.instances)
        insts = (InstanceDesc * max(1, n_inst))()
        for i, ins in enumerate(scene.instances):
            t = np.asarray(ins.transform, dtype=np.float64).reshape(4, 4)
            inv = np.linalg.inv(t) if ins.inverse is None else np.asarray(ins.inverse, dtype=np.float64).reshape(4, 4)
            insts[i].shape = int(ins.shape)
            insts[i].material = int(ins.material)
            insts[i].transform[:] = t.reshape(-1).tolist()
            insts[i].inverse[:] = inv.reshape(-1).tolist()
        n_mat = len(scene.materials)
        mats = (Material * max(1, n_mat))()
        for i, m in enumerate(scene.materials):
            mats[i].type = int(m.type)
            mats[i].color[:] = [float(c) for c in m.color]
            mats[i].intensity = float(m.intensity)
            mats[i].roughness = float(m.roughness)
            mats[i].mix_factor = float(m.mix_factor)
            mats[i].first = int(m.first)
            mats[i].second = int(m.second)
            mats[i].texture = 0 if m.texture is None else int(m.texture) + 1
            mats[i].normal_map = 0 if m.normal_map is None else int(m.normal_map) + 1
        n_pl = len(scene.point_lights)
        pls = (PointLight * max(1, n_pl))()
        for i, (pos, col, inten) in enumerate(scene.point_lights):
            pls[i].pos[:] = [float(x) for x in pos]
            pls[i].color[:] = [float(x) for x in col]
            pls[i].intensity = float(inten)
        n_al = len(scene.area_lights)
        als = (AreaLight * max(1, n_al))()
        for i, (a, b, c, col, inten) in enumerate(scene.area_lights):
            als[i].a[:] = [float(x) for x in a]
            als[i].b[:] = [float(x) for x in b]
            als[i].c[:] = [float(x) for x in c]
            als[i].color[:] = [float(x) for x in col]
            als[i].intensity = float(inten)
        n_dl = len(scene.directional_lights)
        dls = (DirectionalLight * max(1, n_dl))()
        for i, (direction, angle, col, inten) in enumerate(scene.directional_lights):
            dls[i].direction[:] = [float(x) for x in direction]
            dls[i].angle = float(angle)
            dls[i].color[:] = [float(x) for x in col]
            dls[i].intensity = float(inten)
        self._keep += [meshes, insts, mats, pls, als, dls]
        d = SceneDesc()
        d.meshes = meshes
        d.instances = insts
        d.materials = mats
        d.point_lights = pls
        d.area_lights = als
        d.n_meshes, d.n_instances, d.n_materials = n_meshes, n_inst, n_mat
        d.n_point_lights, d.n_area_lights = n_pl, n_al
        d.directional_lights = dls
        d.n_directional_lights = n_dl
        n_tx = len(scene.textures)
        txs = (Texture * max(1, n_tx))()
        for i, (pixels, gamma) in enumerate(scene.textures):
            px = np.ascontiguousarray(pixels, dtype=np.uint8)
            if px.ndim != 3 or px.shape[2] != 4:
                raise ValueError("texture pixels must have shape (height, width, 4)")
            self._keep.append(px)
            txs[i].rgba = px.ctypes.data_as(c_u8_p)
            txs[i].height, txs[i].width = px.shape[0], px.shape[1]
            txs[i].gamma = float(gamma)
        self._keep.append(txs)
        d.textures = txs
        d.n_textures = n_tx
        if scene.env is not None:
            d.has_environment = 1
            d.env_color[:] = [float(x) for x in scene.env[0]]
            d.env_intensity = float(scene.env[1])
        if scene.sky:
            d.has_environment = 3
        if scene.env_map is not None:
            texels = np.ascontiguousarray(scene.env_map[0], dtype=np.float32)
            if texels.ndim != 3 or texels.shape[2] != 4:
                raise ValueError("environment texels must have shape (height, width, 4)")
            self._keep.append(texels)
            d.has_environment = 2
            d.env_map = texels.ctypes.data_as(C.POINTER(C.c_float))
            d.env_map_height, d.env_map_width = texels.shape[0], texels.shape[1]
            d.env_intensity = float(scene.env_map[1])
        self.desc = d


def make_settings(width, height, geometry_aa=4, material_aa=32, max_path_length=10, seed=0,
                  jitter_mode=JITTER_RANDOM, tile=None) -> RenderSettings:
    s = RenderSettings()
    s.width, s.height = int(width), int(height)
    xs, ys, xe, ye = (0, 0, width, height) if tile is None else tile
    s.x_start, s.y_start, s.x_end, s.y_end = int(xs), int(ys), int(xe), int(ye)
    s.geometry_aa, s.material_aa, s.max_path_length = int(geometry_aa), int(material_aa), int(max_path_length)
    s.seed, s.jitter_mode = int(seed), int(jitter_mode)
    return s
