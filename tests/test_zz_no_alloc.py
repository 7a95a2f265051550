"""rtgpu_render_device_static -- the render that allocates nothing -- against rtgpu_render_device, through the C ABI."""
import numpy as np
import pytest

from raytracer_amd import abi, api, scenes

pytestmark = pytest.mark.gpu


def test_static_workspace_render_matches_the_allocating_one_twice_in_a_row():
    """Two calls in a row on one scene, the second with a smaller input (fewer pixels and fewer samples): both images are
    bit-identical to rtgpu_render_device's.  The first frame needs several batches of the 64 MiB workspace, the second
    one; the second call leaves the amount of free device memory where it was.  (The first call on a stream can change it
    without the library allocating anything: the HIP runtime sizes a hardware queue's private-segment memory when a kernel
    that needs more of it than any before is first launched there -- the traversal kernels keep the deep end of their
    stacks in private memory.  The workspace's own address is looked up when the scene is made.)"""
    import torch
    sc, cam_m, yfov = scenes.box_scene(with_mirror=True)
    dev = torch.device("cuda", 0)
    stream = torch.cuda.current_stream()
    r = api.Renderer()
    with api.Scene(sc) as g:
        for k, (W, H, G, M, tile) in enumerate(((352, 264, 3, 4, None), (200, 150, 2, 2, (8, 5, 190, 141)))):
            cam = api.camera_from_transform(scenes.x_fov_degrees(yfov, W / H), cam_m)
            st = abi.make_settings(W, H, geometry_aa=G, material_aa=M, seed=5 + k, tile=tile)
            want = torch.full((H, W, 3), -1.0, dtype=torch.float32, device=dev)
            r.render_device(g, cam, st, want, stream)
            got = torch.full((H, W, 3), -1.0, dtype=torch.float32, device=dev)
            torch.cuda.synchronize()
            free_before = torch.cuda.mem_get_info()[0]
            r.render_device_static(g, cam, st, got, stream)
            torch.cuda.synchronize()
            if k == 1:
                assert torch.cuda.mem_get_info()[0] == free_before
            assert torch.equal(got, want)
            assert float(want[want >= 0].mean()) > 0.01
        # more paths per pixel than the workspace holds: refused with a clear message, nothing launched
        st = abi.make_settings(64, 48, geometry_aa=1024, material_aa=1024)
        with pytest.raises(api.RtgpuError, match="static workspace"):
            r.render_device_static(g, cam, st, got, stream)

