"""rtgpu_render with n_gpus > 1 on a box with one GPU: RTGPU_MULTI_LOOPBACK puts every rank on the scene's own device, so the
scene copies, the host thread per rank, the compact buffers, rank 0's stage layout, the scatter kernel, the set-up's failure
paths and the gather's deadline all execute; only the transfers themselves are device-to-device copies (or, with "rccl",
send / receive pairs of a one-rank communicator) instead of ncclSend / ncclRecv between devices.  The reference's counterpart
is one process per tile plus --mergetiles (exe/Raytracer.cpp:260-277, io/TileFile.cpp:8-70)."""
import os
import time

import numpy as np
import pytest

from raytracer_amd import abi, api, scenes

pytestmark = pytest.mark.gpu


@pytest.fixture
def env():
    """Sets environment variables for one test and restores them afterwards (the library reads them at every call)."""
    saved = {}

    def setenv(**kv):
        for k, v in kv.items():
            saved.setdefault(k, os.environ.get(k))
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = str(v)

    yield setenv
    for k, v in saved.items():
        if v is None:
            os.environ.pop(k, None)
        else:
            os.environ[k] = v


def frame_setup(W=200, H=136, G=2, M=2, tile=None, seed=3):
    sc, cam_m, yfov = scenes.box_scene(with_mirror=True)
    cam = api.camera_from_transform(scenes.x_fov_degrees(yfov, W / H), cam_m)
    return sc, cam, abi.make_settings(W, H, geometry_aa=G, material_aa=M, seed=seed, tile=tile)


def render(g, cam, st, n_gpus, fill=-1.0):
    st = abi.RenderSettings.from_buffer_copy(st)
    st.n_gpus = n_gpus
    fb = np.full((st.height, st.width, 3), fill, dtype=np.float32)
    out, stats = api.Renderer().render(g, cam, st, fb)
    return out, stats


@pytest.mark.parametrize("mode", ["copy", "rccl"])
def test_loopback_ranks_assemble_the_single_pass_frame(env, mode):
    """G = 2, 3, 4, 8 logical ranks (a count that does not divide the squares evenly among them), a tile that cuts squares, and a
    second call on the kept set-up: the same bits as n_gpus = 1, pixels outside the tile untouched, ray counts equal."""
    env(RTGPU_MULTI_LOOPBACK=mode)
    sc, cam, st = frame_setup(tile=(5, 3, 197, 130))
    with api.Scene(sc) as g:
        want, wstats = render(g, cam, st, 1)
        assert np.all(want[0] == -1.0) and float(want[3:130, 5:197].mean()) > 0.01
        for G in (2, 3, 4, 8, 4):
            got, stats = render(g, cam, st, G)
            assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), "G = %d" % G
            assert (stats.primary_rays, stats.extension_rays, stats.shadow_rays) == (wstats.primary_rays, wstats.extension_rays, wstats.shadow_rays)
        # a full frame, more samples: the kept copies and buffers grow
        _, cam2, st2 = frame_setup(W=320, H=200, G=3, M=4, seed=9)
        want2, _ = render(g, cam2, st2, 1)
        got2, _ = render(g, cam2, st2, 4)
        assert np.array_equal(got2.view(np.uint32), want2.view(np.uint32))


def test_gather_deadline_aborts_and_the_next_call_recovers(env):
    """A transfer that does not arrive (test hook: a kernel in front of rank 0's receives that waits 20 s unless released):
    the call gives up after RTGPU_COMM_TIMEOUT_MS with RTGPU_ERR_COMM instead of waiting, releases what it enqueued, drops the
    multi-GPU set-up, and the next call builds it again and renders the right frame."""
    env(RTGPU_MULTI_LOOPBACK="copy")
    sc, cam, st = frame_setup()
    with api.Scene(sc) as g:
        want, _ = render(g, cam, st, 1)
        got, _ = render(g, cam, st, 2)
        assert np.array_equal(got.view(np.uint32), want.view(np.uint32))
        env(RTGPU_MULTI_TEST_STALL_MS=20000, RTGPU_COMM_TIMEOUT_MS=300)
        t0 = time.perf_counter()
        with pytest.raises(api.RtgpuError, match="rtgpu error 9: the gather did not complete within 300 ms"):
            render(g, cam, st, 2)
        assert time.perf_counter() - t0 < 10.0
        env(RTGPU_MULTI_TEST_STALL_MS=None, RTGPU_COMM_TIMEOUT_MS=None)
        got, _ = render(g, cam, st, 2)
        assert np.array_equal(got.view(np.uint32), want.view(np.uint32))
        # a short stall inside the limit is waited for
        env(RTGPU_MULTI_TEST_STALL_MS=200, RTGPU_COMM_TIMEOUT_MS=5000)
        got, _ = render(g, cam, st, 2)
        assert np.array_equal(got.view(np.uint32), want.view(np.uint32))


@pytest.mark.parametrize("step,code", [("replicate", MemoryError), ("stream", api.RtgpuError), ("comm", api.RtgpuError)])
def test_failed_setup_leaves_nothing_half_built(env, step, code):
    """The set-up of a multi-GPU render fails at the scene copy, at a stream or at the communicator (injected): the error is
    reported, nothing is attached to the scene, and the same call succeeds once the cause is gone."""
    env(RTGPU_MULTI_LOOPBACK="copy", RTGPU_MULTI_TEST_FAIL=step)
    sc, cam, st = frame_setup()
    with api.Scene(sc) as g:
        want, _ = render(g, cam, st, 1)
        for _ in range(2):
            with pytest.raises(code, match="injected failure"):
                render(g, cam, st, 3)
        env(RTGPU_MULTI_TEST_FAIL=None)
        got, _ = render(g, cam, st, 3)
        assert np.array_equal(got.view(np.uint32), want.view(np.uint32))


def test_send_without_receive_is_an_error_not_a_hang(env):
    """Through RCCL (one-rank communicator, sends to itself): rank 0 posts no receive for the peers' sends.  RCCL either refuses
    the group or the transfer waits until the deadline aborts the communicator; both end in RTGPU_ERR_COMM within the time
    limit, the set-up is dropped, and the next call renders the right frame."""
    env(RTGPU_MULTI_LOOPBACK="rccl", RTGPU_COMM_TIMEOUT_MS=3000)
    sc, cam, st = frame_setup()
    with api.Scene(sc) as g:
        want, _ = render(g, cam, st, 1)
        env(RTGPU_MULTI_TEST_FAIL="recv")
        t0 = time.perf_counter()
        with pytest.raises(api.RtgpuError, match="rtgpu error 9"):
            render(g, cam, st, 2)
        assert time.perf_counter() - t0 < 30.0
        env(RTGPU_MULTI_TEST_FAIL=None)
        got, _ = render(g, cam, st, 2)
        assert np.array_equal(got.view(np.uint32), want.view(np.uint32))


def test_more_gpus_than_present_is_refused(env):
    env(RTGPU_MULTI_LOOPBACK=None)
    sc, cam, st = frame_setup()
    with api.Scene(sc) as g:
        with pytest.raises(api.RtgpuError, match="GPUs"):
            render(g, cam, st, api.device_count() + 1)
