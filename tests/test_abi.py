"""CPU tests of the drop-in boundary: the library loads, exports every declared symbol, and the
ctypes mirror matches the C structs byte for byte.  No compute call is made (no GPU here)."""
import ctypes as C
import os
import re
import subprocess
import tempfile

import pytest

from raytracer_amd import abi, api

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "rtgpu.h")


def declared_functions():
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(rtgpu_[a-z_]+)\s*\(", text)) - {"rtgpu_progress_cb"})


def test_library_exports_every_declared_symbol():
    lib = api.lib()
    names = declared_functions()
    assert len(names) >= 15
    for n in names:
        assert hasattr(lib, n), n
    assert sorted(api.EXPORTS) == names


def test_struct_layouts_match_the_header():
    structs = {"rtgpu_mesh_desc": abi.MeshDesc, "rtgpu_instance_desc": abi.InstanceDesc, "rtgpu_material": abi.Material,
               "rtgpu_point_light": abi.PointLight, "rtgpu_area_light": abi.AreaLight,
               "rtgpu_directional_light": abi.DirectionalLight, "rtgpu_texture": abi.Texture, "rtgpu_scene_desc": abi.SceneDesc,
               "rtgpu_hit": abi.Hit, "rtgpu_camera": abi.Camera, "rtgpu_render_settings": abi.RenderSettings,
               "rtgpu_render_stats": abi.RenderStats, "rtgpu_scene_info": abi.SceneInfo}
    src = '#include <stdio.h>\n#include "rtgpu.h"\nint main(void){\n'
    for n in structs:
        src += f'printf("{n} %zu\\n", sizeof({n}));\n'
    src += "return 0;}\n"
    with tempfile.TemporaryDirectory() as td:
        c = os.path.join(td, "sz.c")
        open(c, "w").write(src)
        exe = os.path.join(td, "sz")
        subprocess.check_call(["gcc", "-std=c99", "-I", os.path.join(ROOT, "include"), c, "-o", exe])
        out = subprocess.check_output([exe]).decode().split()
    sizes = dict(zip(out[0::2], map(int, out[1::2])))
    for n, t in structs.items():
        assert sizes[n] == C.sizeof(t), n


def test_header_is_plain_c():
    with tempfile.TemporaryDirectory() as td:
        c = os.path.join(td, "x.c")
        open(c, "w").write('#include "rtgpu.h"\nint main(void){return (int)sizeof(rtgpu_hit) - 20;}\n')
        subprocess.check_call(["gcc", "-std=c99", "-pedantic", "-Werror", "-I", os.path.join(ROOT, "include"), c, "-o", os.path.join(td, "x")])


def test_cpp_adapter_compiles():
    """include/rtgpu_adapter.hpp (the Renderer::render / Scene::traceRay shaped C++ front end)."""
    with tempfile.TemporaryDirectory() as td:
        c = os.path.join(td, "a.cpp")
        open(c, "w").write('#include "rtgpu_adapter.hpp"\nint main(){ rtgpu::RenderSettings s; return s.geometryAAModifier - 4; }\n')
        subprocess.check_call(["g++", "-std=c++17", "-Wall", "-Werror", "-fsyntax-only", "-I", os.path.join(ROOT, "include"), c])


def test_argument_errors_without_a_gpu():
    """Argument checks happen before any device work: same failure classes as the reference
    (std::invalid_argument -> ValueError)."""
    lib = api.lib()
    h = C.c_void_p()
    assert lib.rtgpu_scene_create(None, C.byref(h)) == 1
    assert b"null" in lib.rtgpu_last_error()
    d = abi.SceneDesc()
    assert lib.rtgpu_scene_create(C.byref(d), C.byref(h)) == 1  # no instances
    st = abi.make_settings(0, 10)
    cam = abi.Camera()
    assert lib.rtgpu_generate_primary_device(C.byref(cam), C.byref(st), 0, None, None, None) == 1
    st = abi.make_settings(16, 16, tile=(8, 0, 4, 16))
    assert lib.rtgpu_generate_primary_device(C.byref(cam), C.byref(st), 0, None, None, None) == 1
    assert b"Invalid tile bounds" in lib.rtgpu_last_error()  # film/Tile.cpp:11
    assert lib.rtgpu_generate_primary_device(None, C.byref(st), 0, None, None, None) == 5
    assert b"No camera in scene." in lib.rtgpu_last_error()  # renderer/Renderer.cpp:318
    # one pixel's paths must fit the 25 bits a queue entry has for the path: 2^25 is the last count accepted
    st = abi.make_settings(16, 16, geometry_aa=1 << 13, material_aa=(1 << 12) + 1)
    assert lib.rtgpu_generate_primary_device(C.byref(cam), C.byref(st), 0, None, None, None) == 6
    assert b"2^25" in lib.rtgpu_last_error()
    st = abi.make_settings(16, 16, geometry_aa=1 << 13, material_aa=1 << 12)
    assert lib.rtgpu_generate_primary_device(C.byref(cam), C.byref(st), 0, None, None, None) == 1   # accepted; the null buffers are what fails
    assert lib.rtgpu_comm_set_timeout_ms(None, 5) == 1
    # pixel indices stay below 2^31: frames of more than 2^30 pixels are refused
    st = abi.make_settings(1 << 15, (1 << 15) + 1, tile=(0, 0, 16, 16))
    assert lib.rtgpu_generate_primary_device(C.byref(cam), C.byref(st), 0, None, None, None) == 6
    assert b"2^30" in lib.rtgpu_last_error()


def test_packed_scene_validation():
    import numpy as np
    from raytracer_amd import scenes
    sc = scenes.two_spheres_scene()
    abi.PackedScene(sc)
    m = scenes.uv_sphere(8, 4)
    m.indices = m.indices.copy()
    m.indices[0, 0] = 10_000
    sc.meshes = [m]
    with pytest.raises(ValueError):
        abi.PackedScene(sc)


def test_camera_from_transform_matches_oracle():
    import numpy as np
    import oracle
    from raytracer_amd import scenes
    for eye, tgt in [((1, 2, 3), (0, 0, 0)), ((0.5, 0.5, -3), (0.5, 0.5, 0)), ((-4, 1, 2), (3, 2, -1))]:
        m = scenes.look_at(eye, tgt)
        a = api.camera_from_transform(47.0, m)
        b = oracle.camera_from_transform(47.0, m)
        assert bytes(a) == bytes(b)
    with pytest.raises(ValueError):
        bad = np.eye(4)
        bad[:3, 1] = bad[:3, 2]  # up parallel to the view direction -> "The vectors are colinear!"
        api.camera_from_transform(40.0, bad)
