"""ctypes front end of the CPU oracle (librt_oracle.so).

TEST INFRASTRUCTURE ONLY: imported by tests/, by __graft_entry__.smoke() and by the
cpu_baseline leg of bench.py -- never by the product package.  Parity pinning: see rt_oracle.h.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

from raytracer_amd import abi

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "librt_oracle.so")


def build(force=False):
    if force or not os.path.exists(_LIB_PATH):
        subprocess.check_call(["make", "-C", _HERE, "-s"] + (["-B"] if force else []))
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        L = C.CDLL(_LIB_PATH)
        fp, u32p, u8p = abi.c_float_p, abi.c_u32_p, abi.c_u8_p
        L.orc_scene_create.restype = C.c_void_p
        L.orc_scene_create.argtypes = [C.POINTER(abi.SceneDesc), C.c_int]
        L.orc_scene_destroy.argtypes = [C.c_void_p]
        L.orc_build_photon_map.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_uint64, C.c_uint64, C.c_uint32, C.c_int]
        L.orc_build_photon_map.restype = C.c_uint64
        L.orc_get_photons.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64]
        L.orc_get_photons.restype = C.c_uint64
        L.orc_set_photons.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_int32, C.c_int32]
        L.orc_set_photons.restype = None
        L.orc_gather_photons.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_int]
        L.orc_gather_photons.restype = None
        L.orc_trace_closest.argtypes = [C.c_void_p, fp, fp, C.c_uint64, C.c_void_p, u32p, C.c_int]
        L.orc_trace_any.argtypes = [C.c_void_p, fp, fp, fp, C.c_uint64, u8p, C.c_int]
        L.orc_brute_closest.argtypes = [C.c_void_p, fp, fp, C.c_uint64, C.c_void_p, u32p, C.c_int]
        L.orc_render_pixel_probe.restype = C.c_int64
        L.orc_render_pixel_probe.argtypes = [C.c_void_p, C.POINTER(abi.Camera), C.POINTER(abi.RenderSettings), C.c_int, C.c_int, C.c_int64, C.c_int,
                                             fp, C.POINTER(C.c_int)]
        L.orc_attribute_pixel_to_libm.restype = C.c_int
        L.orc_attribute_pixel_to_libm.argtypes = [C.c_void_p, C.POINTER(abi.Camera), C.POINTER(abi.RenderSettings), C.c_int, C.c_int, fp, C.c_float, C.c_float,
                                                  C.POINTER(C.c_int64), C.POINTER(C.c_int), C.POINTER(C.c_int)]
        L.orc_scene_set_gather_rule.argtypes = [C.c_void_p, C.c_int]
        L.orc_scene_set_gather_rule.restype = None
        L.orc_scene_set_closest_rule.argtypes = [C.c_void_p, C.c_int]
        L.orc_scene_set_closest_rule.restype = None
        L.orc_scene_set_primary_rays.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_scene_set_primary_rays.restype = None
        L.orc_ray_hits_instance_box.argtypes = [C.c_void_p, fp, fp, C.c_uint64, C.c_uint32, u8p]
        L.orc_generate_primary.argtypes = [C.POINTER(abi.Camera), C.POINTER(abi.RenderSettings), C.c_int32, fp, fp]
        L.orc_render.argtypes = [C.c_void_p, C.POINTER(abi.Camera), C.POINTER(abi.RenderSettings), fp,
                                 C.POINTER(abi.RenderStats), C.c_int]
        L.orc_triangle_area.restype = C.c_float
        L.orc_triangle_area.argtypes = [fp, fp, fp]
        L.orc_fresnel_mix_factor.restype = C.c_float
        L.orc_fresnel_mix_factor.argtypes = [fp, fp, C.c_float]
        L.orc_aabb_of_transformed.argtypes = [fp, fp, C.POINTER(C.c_double), fp, fp]
        L.orc_aabb_merge.argtypes = [fp, fp, fp, fp, fp, fp]
        L.orc_aabb_area.restype = C.c_double
        L.orc_aabb_area.argtypes = [fp, fp]
        L.orc_rotate_quaternion.argtypes = [C.c_double] * 4 + [C.POINTER(C.c_double), C.POINTER(C.c_double)]
        L.orc_transform_vector.argtypes = [C.POINTER(C.c_double), fp, fp]
        L.orc_transform_point.argtypes = [C.POINTER(C.c_double), fp, fp]
        L.orc_triangle_intersect.restype = C.c_int
        L.orc_triangle_intersect.argtypes = [fp, fp, fp, fp, fp, fp, fp, fp]
        L.orc_sphere_intersect.restype = C.c_int
        L.orc_sphere_intersect.argtypes = [fp, fp, C.POINTER(C.c_double)]
        L.orc_rand_unit.restype = C.c_float
        L.orc_rand_unit.argtypes = [C.c_uint32] * 5
        L.orc_camera_from_transform.argtypes = [C.c_double, C.POINTER(C.c_double), C.POINTER(abi.Camera)]
        L.orc_scene_node_count.restype = C.c_uint64
        L.orc_scene_node_count.argtypes = [C.c_void_p]
        _lib = L
    return _lib


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _fp(a):
    return a.ctypes.data_as(abi.c_float_p)


def default_threads():
    return max(1, min(os.cpu_count() or 1, int(os.environ.get("OMP_NUM_THREADS", os.cpu_count() or 1))))


class OracleScene:
    def __init__(self, scene: abi.SceneData, threads=None):
        self.threads = threads or default_threads()
        self._packed = abi.PackedScene(scene)
        self._h = lib().orc_scene_create(C.byref(self._packed.desc), self.threads)
        if not self._h:
            raise RuntimeError("orc_scene_create failed")

    def close(self):
        if self._h:
            lib().orc_scene_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def node_count(self):
        return int(lib().orc_scene_node_count(self._h))

    def trace_closest(self, o, d, counts=False, threads=None):
        o, d = _f32(o).reshape(-1, 3), _f32(d).reshape(-1, 3)
        n = o.shape[0]
        out = np.empty(n, dtype=abi.HIT_DTYPE)
        cnt = np.zeros((n, 2), dtype=np.uint32) if counts else None
        lib().orc_trace_closest(self._h, _fp(o), _fp(d), n, out.ctypes.data,
                                cnt.ctypes.data_as(abi.c_u32_p) if counts else None, threads or self.threads)
        return (out, cnt) if counts else out

    def trace_any(self, o, d, tmax, threads=None):
        o, d, tmax = _f32(o).reshape(-1, 3), _f32(d).reshape(-1, 3), _f32(tmax).reshape(-1)
        n = o.shape[0]
        out = np.empty(n, dtype=np.uint8)
        lib().orc_trace_any(self._h, _fp(o), _fp(d), _fp(tmax), n, out.ctypes.data_as(abi.c_u8_p), threads or self.threads)
        return out

    def brute_closest(self, o, d, threads=None):
        o, d = _f32(o).reshape(-1, 3), _f32(d).reshape(-1, 3)
        n = o.shape[0]
        out = np.empty(n, dtype=abi.HIT_DTYPE)
        tied = np.zeros(n, dtype=np.uint32)
        lib().orc_brute_closest(self._h, _fp(o), _fp(d), n, out.ctypes.data, tied.ctypes.data_as(abi.c_u32_p), threads or self.threads)
        return out, tied

    def ray_hits_instance_box(self, o, d, inst):
        o, d = _f32(o).reshape(-1, 3), _f32(d).reshape(-1, 3)
        out = np.empty(o.shape[0], dtype=np.uint8)
        lib().orc_ray_hits_instance_box(self._h, _fp(o), _fp(d), o.shape[0], int(inst), out.ctypes.data_as(abi.c_u8_p))
        return out

    # ---- photon map (photonmapping/PhotonMapBuilder.cpp:15-41) ----
    def build_photon_map(self, mode, depth, rays_per_point_light, rays_per_area_light, seed=0, threads=None):
        return int(lib().orc_build_photon_map(self._h, int(mode), int(depth), int(rays_per_point_light), int(rays_per_area_light),
                                              int(seed), threads or self.threads))

    def photons(self):
        n = int(lib().orc_get_photons(self._h, None, 0))
        out = np.empty(n, dtype=abi.PHOTON_DTYPE)
        if n:
            lib().orc_get_photons(self._h, out.ctypes.data_as(C.c_void_p), n)
        return out

    def set_photons(self, photons, mode, depth):
        ph = np.ascontiguousarray(photons, dtype=abi.PHOTON_DTYPE)
        lib().orc_set_photons(self._h, ph.ctypes.data_as(C.c_void_p), len(ph), int(mode), int(depth))

    def gather_photons(self, points, dirs, threads=None):
        p, d = _f32(points).reshape(-1, 3), _f32(dirs).reshape(-1, 3)
        out = np.empty(len(p), dtype=abi.GATHER_DTYPE)
        lib().orc_gather_photons(self._h, p.ctypes.data_as(C.c_void_p), d.ctypes.data_as(C.c_void_p), len(p),
                                 out.ctypes.data_as(C.c_void_p), threads or self.threads)
        return out

    def set_closest_rule(self, rule: int):
        """0: the reference's tree walk (default).  1: the closest hit by the tree-independent definition the GPU kernels
        implement (smallest t over all primitives; equal t -> the smaller (instance, triangle)), by a search without boxes.
        A test hook: it shows that a difference comes from an exact-distance tie, where the reference's pick depends on
        its tree."""
        lib().orc_scene_set_closest_rule(self._h, int(rule))

    def set_gather_rule(self, rule: int):
        """0: the reference's photon search (compares rounded square roots; default).  1: the 20 nearest accepted photons by
        squared distance, every photon looked at -- the set an exact search finds.  A test hook: it shows that a difference
        comes from photons that tie for the 20th place once their distances are rounded."""
        lib().orc_scene_set_gather_rule(self._h, int(rule))

    def set_primary_rays(self, origins=None, dirs=None):
        """Camera rays for later renders from arrays shaped (geometry_aa, height, width, 3) instead of the oracle's own
        camera (None restores it).  A test hook: it shows that a difference comes from the camera rays' last bits."""
        if origins is None:
            self._primary = None
            lib().orc_scene_set_primary_rays(self._h, None, None)
            return
        o, d = np.ascontiguousarray(origins, dtype=np.float32), np.ascontiguousarray(dirs, dtype=np.float32)
        assert o.shape == d.shape and o.ndim == 4 and o.shape[3] == 3
        self._primary = (o, d)      # kept alive while the scene uses them
        lib().orc_scene_set_primary_rays(self._h, o.ctypes.data, d.ctypes.data)

    LIBM_NAMES = ("acos", "asin", "atan2", "tan", "sin", "cos", "pow")

    def render_pixel_probe(self, cam, st, x, y, target=-1, ulps=0):
        """Pixel (x, y) with the result of its `target`-th libm call moved by `ulps` units in the last place: (rgb, number of
        libm calls the pixel made)."""
        rgb = np.zeros(3, dtype=np.float32)
        n = lib().orc_render_pixel_probe(self._h, C.byref(cam), C.byref(st), int(x), int(y), int(target), int(ulps), _fp(rgb), None)
        return rgb, int(n)

    def attribute_pixel_to_libm(self, cam, st, x, y, want, tol_rel=1e-4, floor=0.0):
        """Looks for one libm call of pixel (x, y) whose result, moved by one unit in the last place, turns the oracle's pixel
        into `want`: (call index, +1 / -1, function name) or None."""
        w = np.ascontiguousarray(want, dtype=np.float32)
        call, ulps, fn = C.c_int64(), C.c_int(), C.c_int()
        ok = lib().orc_attribute_pixel_to_libm(self._h, C.byref(cam), C.byref(st), int(x), int(y), _fp(w), float(tol_rel), float(floor),
                                               C.byref(call), C.byref(ulps), C.byref(fn))
        return (int(call.value), int(ulps.value), self.LIBM_NAMES[fn.value] if 0 <= fn.value < len(self.LIBM_NAMES) else "?") if ok else None

    def render(self, cam: abi.Camera, st: abi.RenderSettings, threads=None):
        img = np.zeros((st.height, st.width, 3), dtype=np.float32)
        stats = abi.RenderStats()
        lib().orc_render(self._h, C.byref(cam), C.byref(st), _fp(img), C.byref(stats), threads or self.threads)
        return img, stats


def camera_from_transform(fov_x_degrees, transform):
    cam = abi.Camera()
    m = (C.c_double * 16)(*np.asarray(transform, dtype=np.float64).reshape(-1).tolist())
    lib().orc_camera_from_transform(float(fov_x_degrees), m, C.byref(cam))
    return cam


def generate_primary(cam, st, sample_index=0):
    n = (st.x_end - st.x_start) * (st.y_end - st.y_start)
    o = np.empty((n, 3), dtype=np.float32)
    d = np.empty((n, 3), dtype=np.float32)
    lib().orc_generate_primary(C.byref(cam), C.byref(st), int(sample_index), _fp(o), _fp(d))
    return o, d


def kd_nearest_explicit(points, child0, child1, axis, target, count, max_radius=None):
    """KDTree::getElementsNearestTo on a hand-built tree (node i holds point i, root = node 0): returns
    (found, max_dist, point indices nearest first with -1 for empty entries)."""
    L = lib()
    L.orc_kd_nearest_explicit.restype = C.c_uint32
    L.orc_kd_nearest_explicit.argtypes = [abi.c_float_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, abi.c_float_p,
                                          C.c_uint32, C.c_float, C.c_void_p, C.POINTER(C.c_float)]
    p = _f32(points).reshape(-1, 3)
    c0, c1, ax = (np.ascontiguousarray(a, dtype=np.int32) for a in (child0, child1, axis))
    t = _f32(target).reshape(3)
    res = np.full(count, -1, dtype=np.int64)
    md = C.c_float()
    r = np.finfo(np.float32).max if max_radius is None else max_radius
    found = L.orc_kd_nearest_explicit(_fp(p), c0.ctypes.data, c1.ctypes.data, ax.ctypes.data, len(p), _fp(t), int(count),
                                      float(r), res.ctypes.data, C.byref(md))
    return int(found), np.float32(md.value), res


def kd_nearest_built(points, target, count, max_radius=None):
    """The same query on the tree KDTreeBuilder::build makes over `points`."""
    L = lib()
    L.orc_kd_nearest_built.restype = C.c_uint32
    L.orc_kd_nearest_built.argtypes = [abi.c_float_p, C.c_uint32, abi.c_float_p, C.c_uint32, C.c_float, C.c_void_p, C.POINTER(C.c_float)]
    p = _f32(points).reshape(-1, 3)
    t = _f32(target).reshape(3)
    res = np.full(count, -1, dtype=np.int64)
    md = C.c_float()
    r = np.finfo(np.float32).max if max_radius is None else max_radius
    found = L.orc_kd_nearest_built(_fp(p), len(p), _fp(t), int(count), float(r), res.ctypes.data, C.byref(md))
    return int(found), np.float32(md.value), res


def aabb_corners(lo, hi):
    L = lib()
    L.orc_aabb_corners.argtypes = [abi.c_float_p, abi.c_float_p, abi.c_float_p]
    out = np.empty((8, 3), dtype=np.float32)
    L.orc_aabb_corners(_fp(_f32(lo).reshape(3)), _fp(_f32(hi).reshape(3)), _fp(out))
    return out
