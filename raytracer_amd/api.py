"""Host-side mirror of the reference's query/render interface on top of the C ABI (librtgpu.so).

`Scene.trace_ray(s)` / `Scene.test_visibility` mirror Scene::traceRay / Scene::testVisibility
(raytracer/src/lib/scene/renderable/Scene.h:96-100) over ray batches and `Renderer.render`
mirrors Renderer::render (raytracer/src/lib/renderer/Renderer.h:17).  Errors of the C ABI are
raised as the exception types the reference throws (std::invalid_argument -> ValueError,
std::runtime_error -> RuntimeError).

There is no CPU path: importing works without a GPU (so the ABI can be inspected), but every
compute call needs librtgpu.so and a gfx950 device and fails loudly otherwise.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

from . import abi

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("RTGPU_LIB") or os.path.join(_HERE, "librtgpu.so")

EXPORTS = [
    "rtgpu_last_error", "rtgpu_device_count", "rtgpu_set_device", "rtgpu_scene_create", "rtgpu_scene_destroy",
    "rtgpu_trace_closest", "rtgpu_trace_closest_device", "rtgpu_trace_any", "rtgpu_trace_any_device",
    "rtgpu_generate_primary_device", "rtgpu_render", "rtgpu_render_device", "rtgpu_camera_from_transform",
    "rtgpu_scene_get_info", "rtgpu_trace_closest_count_device",
    "rtgpu_gltf_load", "rtgpu_gltf_set_workdir", "rtgpu_gltf_scene_desc", "rtgpu_gltf_camera", "rtgpu_gltf_camera_count", "rtgpu_gltf_free",
    "rtgpu_write_exr", "rtgpu_write_png",
    "rtgpu_scene_build_photon_map", "rtgpu_scene_get_photons", "rtgpu_scene_set_photons", "rtgpu_scene_gather_photons",
    "rtgpu_comm_unique_id", "rtgpu_comm_create", "rtgpu_comm_destroy", "rtgpu_comm_set_timeout_ms", "rtgpu_tile_buffer_floats", "rtgpu_render_tiles_device",
    "rtgpu_gather_tiles_device", "rtgpu_scatter_tiles_device", "rtgpu_scene_replicate", "rtgpu_render_work_device", "rtgpu_render_profile_device", "rtgpu_render_device_static",
    "rtgpu_host_alloc", "rtgpu_host_free",
]

_lib = None


class RtgpuError(RuntimeError):
    pass


def lib():
    """Loads librtgpu.so (built in-tree by `make -C raytracer_amd/csrc`)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RtgpuError(f"{LIB_PATH} is missing: build it with `make -C raytracer_amd/csrc` "
                         "(there is no CPU fallback for the ray-tracing path)")
    # PyTorch wheels bundle their own libamdhip64/libhsa-runtime64.  Two HIP runtimes in one process
    # do not both see the GPU, so when torch is installed it is imported first and librtgpu.so
    # binds to the runtime torch brought (same SONAME); without torch the system ROCm one is used.
    if os.environ.get("RTGPU_SKIP_TORCH_PRELOAD") != "1":
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
    L = C.CDLL(LIB_PATH)
    fp, u32p, u8p, vp = abi.c_float_p, abi.c_u32_p, abi.c_u8_p, C.c_void_p
    L.rtgpu_last_error.restype = C.c_char_p
    L.rtgpu_device_count.restype = C.c_int
    L.rtgpu_set_device.argtypes = [C.c_int]
    L.rtgpu_scene_create.argtypes = [C.POINTER(abi.SceneDesc), C.POINTER(vp)]
    L.rtgpu_scene_destroy.argtypes = [vp]
    L.rtgpu_scene_destroy.restype = None
    L.rtgpu_trace_closest.argtypes = [vp, fp, fp, C.c_uint64, vp]
    L.rtgpu_host_alloc.argtypes = [C.POINTER(vp), C.c_uint64]
    L.rtgpu_host_free.argtypes = [vp]
    L.rtgpu_trace_closest_device.argtypes = [vp, vp, vp, C.c_uint64, vp, vp]
    L.rtgpu_trace_any.argtypes = [vp, fp, fp, fp, C.c_uint64, u8p]
    L.rtgpu_trace_any_device.argtypes = [vp, vp, vp, vp, C.c_uint64, vp, vp]
    L.rtgpu_generate_primary_device.argtypes = [C.POINTER(abi.Camera), C.POINTER(abi.RenderSettings), C.c_int32, vp, vp, vp]
    L.rtgpu_render.argtypes = [vp, C.POINTER(abi.Camera), C.POINTER(abi.RenderSettings), fp,
                               C.POINTER(abi.RenderStats), abi.PROGRESS_CB, vp]
    L.rtgpu_render_device.argtypes = [vp, C.POINTER(abi.Camera), C.POINTER(abi.RenderSettings), vp,
                                      C.POINTER(abi.RenderStats), vp]
    L.rtgpu_camera_from_transform.argtypes = [C.c_double, C.POINTER(C.c_double), C.POINTER(abi.Camera)]
    L.rtgpu_scene_get_info.argtypes = [vp, C.POINTER(abi.SceneInfo)]
    L.rtgpu_trace_closest_count_device.argtypes = [vp, vp, vp, C.c_uint64, vp, vp]
    L.rtgpu_scene_build_photon_map.argtypes = [vp, C.c_int32, C.c_int32, C.c_uint64, C.c_uint64, C.c_uint32, C.POINTER(C.c_uint64)]
    L.rtgpu_scene_get_photons.argtypes = [vp, vp, C.c_uint64, C.POINTER(C.c_uint64)]
    L.rtgpu_scene_set_photons.argtypes = [vp, vp, C.c_uint64, C.c_int32, C.c_int32]
    L.rtgpu_scene_gather_photons.argtypes = [vp, fp, fp, C.c_uint64, vp]
    L.rtgpu_write_exr.argtypes = [C.c_char_p, fp, C.c_int32, C.c_int32, C.c_int]
    L.rtgpu_write_png.argtypes = [C.c_char_p, fp, C.c_int32, C.c_int32, C.c_double, C.c_double]
    L.rtgpu_comm_unique_id.argtypes = [vp]
    L.rtgpu_comm_create.argtypes = [vp, C.c_int, C.c_int, C.POINTER(vp)]
    L.rtgpu_comm_destroy.argtypes = [vp]
    L.rtgpu_comm_destroy.restype = None
    L.rtgpu_comm_set_timeout_ms.argtypes = [vp, C.c_int]
    L.rtgpu_tile_buffer_floats.argtypes = [C.POINTER(abi.RenderSettings), C.c_int, C.c_int, C.c_int]
    L.rtgpu_tile_buffer_floats.restype = C.c_uint64
    L.rtgpu_render_tiles_device.argtypes = [vp, C.POINTER(abi.Camera), C.POINTER(abi.RenderSettings), C.c_int, C.c_int, C.c_int, vp,
                                            C.POINTER(abi.RenderStats), vp]
    L.rtgpu_gather_tiles_device.argtypes = [vp, C.POINTER(abi.RenderSettings), C.c_int, vp, vp, vp]
    L.rtgpu_scatter_tiles_device.argtypes = [C.POINTER(abi.RenderSettings), C.c_int, C.c_int, vp, vp, vp]
    L.rtgpu_scene_replicate.argtypes = [vp, C.c_int, C.POINTER(vp)]
    L.rtgpu_render_profile_device.argtypes = [vp, C.POINTER(abi.Camera), C.POINTER(abi.RenderSettings), vp, C.POINTER(abi.RenderProfile), vp]
    L.rtgpu_render_device_static.argtypes = [vp, C.POINTER(abi.Camera), C.POINTER(abi.RenderSettings), vp, vp]
    L.rtgpu_render_work_device.argtypes = [vp, C.POINTER(abi.Camera), C.POINTER(abi.RenderSettings), vp, C.POINTER(abi.RenderWork), vp]
    _lib = L
    return L


def _check(rc):
    if rc == 0:
        return
    msg = lib().rtgpu_last_error().decode("utf-8", "replace")
    if rc == 1:
        raise ValueError(msg)              # std::invalid_argument
    if rc == 4:
        raise MemoryError(msg)
    raise RtgpuError(f"rtgpu error {rc}: {msg}")  # std::runtime_error


def device_count() -> int:
    return int(lib().rtgpu_device_count())


def set_device(device: int):
    _check(lib().rtgpu_set_device(int(device)))


def camera_from_transform(fov_x_degrees: float, transform) -> abi.Camera:
    """PerspectiveCamera(fov) + setTransform (camera/PerspectiveCamera.cpp:7-19)."""
    cam = abi.Camera()
    m = (C.c_double * 16)(*np.asarray(transform, dtype=np.float64).reshape(-1).tolist())
    _check(lib().rtgpu_camera_from_transform(float(fov_x_degrees), m, C.byref(cam)))
    return cam


def write_exr(path, frame, full_floats=False):
    """write_to_exr_file (io/EXRWriter.cpp:6-72); frame: (height, width, 3) float32."""
    f = np.ascontiguousarray(frame, dtype=np.float32)
    _check(lib().rtgpu_write_exr(str(path).encode(), f.ctypes.data_as(abi.c_float_p), f.shape[1], f.shape[0], 1 if full_floats else 0))


def write_png(path, frame, exposure=-2.5, gamma=2.2):
    """write_to_png_file (io/PNG.cpp:7-28) with the executable's default exposure and gamma."""
    f = np.ascontiguousarray(frame, dtype=np.float32)
    _check(lib().rtgpu_write_png(str(path).encode(), f.ctypes.data_as(abi.c_float_p), f.shape[1], f.shape[0], float(exposure), float(gamma)))


def _f32(a, cols=3):
    a = np.ascontiguousarray(a, dtype=np.float32)
    return a.reshape(-1, cols) if cols else a.reshape(-1)


def _ptr(t):
    """Device pointer of a torch tensor (or a raw int)."""
    return C.c_void_p(int(t) if isinstance(t, int) else int(t.data_ptr()))


def _stream_ptr(stream):
    if stream is None:
        return C.c_void_p(0)
    return C.c_void_p(int(stream if isinstance(stream, int) else stream.cuda_stream))


class Scene:
    """A built, GPU-resident scene (the reference's Scene after DynamicScene::build)."""

    def __init__(self, scene, _handle=None):
        """scene: abi.SceneData, or an abi.SceneDesc whose arrays the caller keeps alive during this call
        (e.g. gltf.GltfScene.desc)."""
        if _handle is not None:
            self._h = _handle
            return
        h = C.c_void_p()
        if isinstance(scene, abi.SceneDesc):
            _check(lib().rtgpu_scene_create(C.byref(scene), C.byref(h)))
        else:
            packed = abi.PackedScene(scene)
            _check(lib().rtgpu_scene_create(C.byref(packed.desc), C.byref(h)))  # the library copies everything it needs
        self._h = h

    def close(self):
        if getattr(self, "_h", None):
            lib().rtgpu_scene_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    @property
    def handle(self):
        return self._h

    def replicate(self, device: int) -> "Scene":
        """A copy of this built scene on another device of the process (device-to-device, no rebuild)."""
        h = C.c_void_p()
        _check(lib().rtgpu_scene_replicate(self._h, int(device), C.byref(h)))
        return Scene(None, _handle=h)

    def info(self) -> abi.SceneInfo:
        out = abi.SceneInfo()
        _check(lib().rtgpu_scene_get_info(self._h, C.byref(out)))
        return out

    # ---- photon map (photonmapping/PhotonMapBuilder.cpp:15-41, Scene::setPhotonMap) -------------
    def build_photon_map(self, mode, depth, rays_per_point_light, rays_per_area_light, seed=0) -> int:
        """Traces photons from every point and area light, builds the search structure and makes later renders
        of this scene use it (the executable's --pmmode / --pmdepth / --pmrayspointlamp / --pmraysarealamp)."""
        n = C.c_uint64()
        _check(lib().rtgpu_scene_build_photon_map(self._h, int(mode), int(depth), int(rays_per_point_light), int(rays_per_area_light),
                                                  int(seed), C.byref(n)))
        return int(n.value)

    def photons(self) -> np.ndarray:
        n = C.c_uint64()
        _check(lib().rtgpu_scene_get_photons(self._h, None, 0, C.byref(n)))
        out = np.empty(int(n.value), dtype=abi.PHOTON_DTYPE)
        if len(out):
            _check(lib().rtgpu_scene_get_photons(self._h, out.ctypes.data_as(C.c_void_p), len(out), C.byref(n)))
        return out

    def set_photons(self, photons, mode, depth):
        ph = np.ascontiguousarray(photons, dtype=abi.PHOTON_DTYPE)
        _check(lib().rtgpu_scene_set_photons(self._h, ph.ctypes.data_as(C.c_void_p), len(ph), int(mode), int(depth)))

    def gather_photons(self, points, dirs) -> np.ndarray:
        """KDTree::getElementsNearestTo as the diffuse radiance estimate calls it (20 nearest, normal filter)."""
        p, d = _f32(points), _f32(dirs)
        out = np.empty(len(p), dtype=abi.GATHER_DTYPE)
        _check(lib().rtgpu_scene_gather_photons(self._h, p.ctypes.data_as(abi.c_float_p), d.ctypes.data_as(abi.c_float_p), len(p),
                                                out.ctypes.data_as(C.c_void_p)))
        return out

    # ---- Scene::traceRay / traceRays ---------------------------------------------------
    def trace_rays(self, origins, dirs, out=None) -> np.ndarray:
        """Closest hits of a batch of rays in host arrays.  `out` (a C-contiguous HIT_DTYPE array of the batch's length)
        is filled and returned when given: a caller that keeps its result array pays no page faults for a fresh one."""
        o, d = _f32(origins), _f32(dirs)
        if o.shape != d.shape:
            raise ValueError("origins and directions must have the same shape")
        if out is None:
            out = np.empty(o.shape[0], dtype=abi.HIT_DTYPE)
        elif out.dtype != abi.HIT_DTYPE or out.shape != (o.shape[0],) or not out.flags.c_contiguous:
            raise ValueError("out must be a C-contiguous HIT_DTYPE array with one record per ray")
        _check(lib().rtgpu_trace_closest(self._h, o.ctypes.data_as(abi.c_float_p), d.ctypes.data_as(abi.c_float_p),
                                         o.shape[0], C.c_void_p(out.ctypes.data)))
        return out

    def trace_ray(self, origin, direction):
        h = self.trace_rays(np.asarray(origin).reshape(1, 3), np.asarray(direction).reshape(1, 3))[0]
        return None if h["inst"] == abi.INVALID_ID else h  # std::optional<SceneRayHitInfo>

    def trace_rays_device(self, d_origins, d_dirs, n, d_hits, stream=None):
        _check(lib().rtgpu_trace_closest_device(self._h, _ptr(d_origins), _ptr(d_dirs), int(n), _ptr(d_hits), _stream_ptr(stream)))

    def trace_counts_device(self, d_origins, d_dirs, n, d_counts, stream=None):
        _check(lib().rtgpu_trace_closest_count_device(self._h, _ptr(d_origins), _ptr(d_dirs), int(n), _ptr(d_counts), _stream_ptr(stream)))

    # ---- Scene::testVisibility ---------------------------------------------------------
    def test_visibility(self, origins, dirs, max_t, out=None) -> np.ndarray:
        o, d = _f32(origins), _f32(dirs)
        t = _f32(np.broadcast_to(np.asarray(max_t, dtype=np.float32), (o.shape[0],)), cols=0)
        if o.shape != d.shape:
            raise ValueError("origins and directions must have the same shape")
        if out is None:
            out = np.empty(o.shape[0], dtype=np.uint8)
        elif out.dtype != np.uint8 or out.shape != (o.shape[0],) or not out.flags.c_contiguous:
            raise ValueError("out must be a C-contiguous uint8 array with one entry per ray")
        _check(lib().rtgpu_trace_any(self._h, o.ctypes.data_as(abi.c_float_p), d.ctypes.data_as(abi.c_float_p),
                                     t.ctypes.data_as(abi.c_float_p), o.shape[0], out.ctypes.data_as(abi.c_u8_p)))
        return out

    def test_visibility_device(self, d_origins, d_dirs, d_tmax, n, d_occluded, stream=None):
        _check(lib().rtgpu_trace_any_device(self._h, _ptr(d_origins), _ptr(d_dirs), _ptr(d_tmax), int(n), _ptr(d_occluded), _stream_ptr(stream)))


class _HostBlock:
    """Owner of one rtgpu_host_alloc block; the arrays made from it keep it alive."""

    def __init__(self, nbytes):
        p = C.c_void_p()
        _check(lib().rtgpu_host_alloc(C.byref(p), int(nbytes)))
        self.ptr = p.value
        self.nbytes = int(nbytes)
        self._free = lib().rtgpu_host_free      # kept here: module globals may be gone when the last block dies at exit
        self.__array_interface__ = {"shape": (self.nbytes,), "typestr": "|u1", "data": (self.ptr, False), "version": 3}

    def __del__(self):
        if getattr(self, "ptr", None):
            self._free(self.ptr)
            self.ptr = None


def host_empty(shape, dtype=np.float32) -> np.ndarray:
    """A C-contiguous numpy array in page-locked host memory (rtgpu_host_alloc); freed when the last view of it goes.
    Scene.trace_rays / test_visibility move such arrays across PCIe without staging copies."""
    dt = np.dtype(dtype)
    shape = (int(shape),) if np.isscalar(shape) else tuple(int(x) for x in shape)
    n = int(np.prod(shape, dtype=np.int64))
    block = _HostBlock(max(1, n * dt.itemsize))
    return np.asarray(block)[: n * dt.itemsize].view(dt).reshape(shape)


def generate_primary_device(cam: abi.Camera, st: abi.RenderSettings, sample_index, d_origins, d_dirs, stream=None):
    _check(lib().rtgpu_generate_primary_device(C.byref(cam), C.byref(st), int(sample_index), _ptr(d_origins), _ptr(d_dirs), _stream_ptr(stream)))


class Renderer:
    """Renderer::render (renderer/Renderer.cpp:273-304)."""

    def render(self, scene: Scene, cam: abi.Camera, settings: abi.RenderSettings, framebuffer=None, progress=None):
        """Renders the tile of `settings` into `framebuffer` ((H, W, 3) float32, created when None).
        Returns (framebuffer, RenderStats)."""
        if framebuffer is None:
            framebuffer = np.zeros((settings.height, settings.width, 3), dtype=np.float32)
        if framebuffer.dtype != np.float32 or framebuffer.shape != (settings.height, settings.width, 3) or not framebuffer.flags.c_contiguous:
            raise ValueError("framebuffer must be a C-contiguous (H, W, 3) float32 array")
        stats = abi.RenderStats()
        cb = abi.PROGRESS_CB(lambda job, frac, fin, user: progress(job.decode(), frac, bool(fin))) if progress else abi.PROGRESS_CB()
        _check(lib().rtgpu_render(scene.handle, C.byref(cam), C.byref(settings), framebuffer.ctypes.data_as(abi.c_float_p),
                                  C.byref(stats), cb, None))
        return framebuffer, stats

    def render_device(self, scene: Scene, cam: abi.Camera, settings: abi.RenderSettings, d_framebuffer, stream=None):
        stats = abi.RenderStats()
        _check(lib().rtgpu_render_device(scene.handle, C.byref(cam), C.byref(settings), _ptr(d_framebuffer), C.byref(stats), _stream_ptr(stream)))
        return stats

    def render_work_device(self, scene: Scene, cam: abi.Camera, settings: abi.RenderSettings, d_framebuffer, stream=None) -> abi.RenderWork:
        """The same frame with every ray traversed on its own and its boxes / triangles counted (roofline model input)."""
        work = abi.RenderWork()
        _check(lib().rtgpu_render_work_device(scene.handle, C.byref(cam), C.byref(settings), _ptr(d_framebuffer), C.byref(work), _stream_ptr(stream)))
        return work

    def render_device_static(self, scene: Scene, cam: abi.Camera, settings: abi.RenderSettings, d_framebuffer, stream=None):
        """rtgpu_render_device_static: enqueues the render on `stream` using the library's fixed workspace; allocates nothing and
        does not wait."""
        _check(lib().rtgpu_render_device_static(scene.handle, C.byref(cam), C.byref(settings), _ptr(d_framebuffer), _stream_ptr(stream)))

    def render_profile_device(self, scene: Scene, cam: abi.Camera, settings: abi.RenderSettings, d_framebuffer, stream=None) -> abi.RenderProfile:
        """The same frame, one batch at a time with HIP events around every launch: device time per stage."""
        prof = abi.RenderProfile()
        _check(lib().rtgpu_render_profile_device(scene.handle, C.byref(cam), C.byref(settings), _ptr(d_framebuffer), C.byref(prof), _stream_ptr(stream)))
        return prof

    def render_tiles_device(self, scene: Scene, cam: abi.Camera, settings: abi.RenderSettings, tile_size, rank, world, d_tiles, stream=None):
        """This rank's squares of the frame into its compact device buffer (see raytracer_amd.distributed)."""
        stats = abi.RenderStats()
        _check(lib().rtgpu_render_tiles_device(scene.handle, C.byref(cam), C.byref(settings), int(tile_size), int(rank), int(world),
                                               _ptr(d_tiles), C.byref(stats), _stream_ptr(stream)))
        return stats


def tile_buffer_floats(settings: abi.RenderSettings, tile_size, rank, world) -> int:
    return int(lib().rtgpu_tile_buffer_floats(C.byref(settings), int(tile_size), int(rank), int(world)))


def scatter_tiles_device(settings: abi.RenderSettings, tile_size, world, d_all_tiles, d_frame, stream=None):
    _check(lib().rtgpu_scatter_tiles_device(C.byref(settings), int(tile_size), int(world), _ptr(d_all_tiles), _ptr(d_frame), _stream_ptr(stream)))


COMM_ID_BYTES = 128


def comm_unique_id() -> bytes:
    """The id rank 0 makes and the host program hands to the other ranks (ncclGetUniqueId)."""
    buf = (C.c_uint8 * COMM_ID_BYTES)()
    _check(lib().rtgpu_comm_unique_id(buf))
    return bytes(buf)


class Comm:
    """One rank's communicator of the frame gather (RCCL), on the device current when it is made."""

    def __init__(self, unique_id: bytes, rank: int, world: int):
        if len(unique_id) != COMM_ID_BYTES:
            raise ValueError("communicator id must be %d bytes" % COMM_ID_BYTES)
        h = C.c_void_p()
        buf = (C.c_uint8 * COMM_ID_BYTES).from_buffer_copy(unique_id)
        _check(lib().rtgpu_comm_create(buf, int(rank), int(world), C.byref(h)))
        self._h, self.rank, self.world = h, int(rank), int(world)

    def gather_tiles_device(self, settings: abi.RenderSettings, tile_size, d_tiles, d_frame, stream=None):
        _check(lib().rtgpu_gather_tiles_device(self._h, C.byref(settings), int(tile_size), _ptr(d_tiles),
                                               _ptr(d_frame) if d_frame is not None else C.c_void_p(0), _stream_ptr(stream)))

    def set_timeout_ms(self, timeout_ms: int):
        """How long a gather may wait for its transfers before the communicator is aborted (0: the gather only enqueues)."""
        _check(lib().rtgpu_comm_set_timeout_ms(self._h, int(timeout_ms)))

    def close(self):
        if getattr(self, "_h", None):
            lib().rtgpu_comm_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
