"""One frame over the ranks of a node: the square order of the library's multi-GPU path and the frame gather.

The reference scales out with separate processes that render disjoint tiles
(--xstart/--xend/--ystart/--yend, exe/Raytracer.cpp:65-68), write TILE files (io/TileFile.cpp:8-27)
and paste them together with --mergetiles (exe/Raytracer.cpp:260-277, io/TileFile.cpp:29-70).  Here
the settings' tile is cut into T x T squares numbered row by row, square k belongs to rank k % world,
every rank (one process per GPU, scene replicated) renders its squares into a compact buffer
(rtgpu_render_tiles_device) and the buffers meet on rank 0:

* natively, through the library's RCCL send / receive pairs over xGMI and its scatter kernel
  (api.Comm.gather_tiles_device) -- the measured path;
* through a torch.distributed gather of equal padded buffers ("gloo" on host copies: the CPU
  rehearsals of the tests, and the stand-in when the native communicator cannot be made).

Pixels are keyed by their frame position in the RNG, so the assembled frame is bit-identical for
every world size.  The functions on numpy arrays below are the host-side model of the square order
(what rt_multi.hip computes on the device); the tests hold one against the other.
"""
from __future__ import annotations

from typing import List, Tuple

import numpy as np

TILE_SIZE = 32


# ---- the square order (host model) -------------------------------------------------------------

def tile_grid(settings, tile_size: int = TILE_SIZE) -> Tuple[int, int]:
    """(squares per row, squares in all) over the settings' tile."""
    tw, th = settings.x_end - settings.x_start, settings.y_end - settings.y_start
    tiles_x = -(-tw // tile_size)
    return tiles_x, tiles_x * -(-th // tile_size)


def rank_tile_count(n_tiles: int, rank: int, world: int) -> int:
    return (n_tiles - rank + world - 1) // world if n_tiles > rank else 0


def tile_buffer_floats(settings, tile_size: int, rank: int, world: int) -> int:
    """Floats of a rank's compact buffer (mirror of rtgpu_tile_buffer_floats)."""
    _, n_tiles = tile_grid(settings, tile_size)
    return rank_tile_count(n_tiles, rank, world) * tile_size * tile_size * 3


def compact_from_frame(frame: np.ndarray, settings, tile_size: int, rank: int, world: int) -> np.ndarray:
    """The compact buffer rank `rank` would hold if it rendered `frame` ((H, W, 3)): its squares one after the
    other, each row by row, pixels outside the tile zero."""
    T = tile_size
    tiles_x, n_tiles = tile_grid(settings, T)
    out = np.zeros((rank_tile_count(n_tiles, rank, world), T, T, 3), dtype=np.float32)
    for j in range(out.shape[0]):
        k = j * world + rank
        x0, y0 = settings.x_start + (k % tiles_x) * T, settings.y_start + (k // tiles_x) * T
        x1, y1 = min(x0 + T, settings.x_end), min(y0 + T, settings.y_end)
        out[j, :y1 - y0, :x1 - x0] = frame[y0:y1, x0:x1]
    return out.reshape(-1)


def scatter_to_frame(buffers: List[np.ndarray], settings, tile_size: int, frame: np.ndarray = None) -> np.ndarray:
    """Rank 0's last step: the ranks' compact buffers written into FrameBuffer order (pixels outside the
    settings' tile keep what `frame` held)."""
    T, world = tile_size, len(buffers)
    tiles_x, n_tiles = tile_grid(settings, T)
    if frame is None:
        frame = np.zeros((settings.height, settings.width, 3), dtype=np.float32)
    for rank, buf in enumerate(buffers):
        sq = np.asarray(buf, dtype=np.float32).reshape(-1, T, T, 3)
        for j in range(rank_tile_count(n_tiles, rank, world)):
            k = j * world + rank
            x0, y0 = settings.x_start + (k % tiles_x) * T, settings.y_start + (k // tiles_x) * T
            x1, y1 = min(x0 + T, settings.x_end), min(y0 + T, settings.y_end)
            frame[y0:y1, x0:x1] = sq[j, :y1 - y0, :x1 - x0]
    return frame


# ---- the gather ------------------------------------------------------------------------------------

def gather_compact(local, settings, tile_size: int, rank: int, world: int, group=None):
    """torch.distributed gather of the ranks' compact buffers (padded to rank 0's size, the largest) to rank 0 and the
    scatter into frame order there.  `local`: 1-D float32 torch tensor (CPU or GPU).  Returns the (H, W, 3) frame on rank 0
    (a device tensor when the exchange ran on device tensors, else a numpy array), None elsewhere.  The rehearsal / stand-in path: the measured one is api.Comm."""
    import torch
    import torch.distributed as dist

    stride = tile_buffer_floats(settings, tile_size, 0, world)
    if world > 1 and local.is_cuda and dist.get_backend(group) != "gloo":
        # device tensors end to end: the ranks' buffers land side by side and the library's scatter kernel writes the frame
        from . import api
        padded = torch.zeros(stride, dtype=torch.float32, device=local.device)
        padded[:local.numel()] = local
        big = torch.empty(world * stride, dtype=torch.float32, device=local.device) if rank == 0 else None
        dist.gather(padded, list(big.view(world, stride).unbind(0)) if rank == 0 else None, dst=0, group=group)
        if rank != 0:
            return None
        frame = torch.zeros((settings.height, settings.width, 3), dtype=torch.float32, device=local.device)
        api.scatter_tiles_device(settings, tile_size, world, big, frame, torch.cuda.current_stream())
        return frame
    if world == 1:
        parts = [local]
    else:
        padded = torch.zeros(stride, dtype=torch.float32, device=local.device)
        padded[:local.numel()] = local
        if padded.is_cuda and dist.get_backend(group) == "gloo":
            padded = padded.cpu()
        parts = [torch.empty_like(padded) for _ in range(world)] if rank == 0 else None
        dist.gather(padded, parts, dst=0, group=group)
    if rank != 0:
        return None
    bufs = [p.detach().cpu().numpy()[:tile_buffer_floats(settings, tile_size, r, world)] for r, p in enumerate(parts)]
    return scatter_to_frame(bufs, settings, tile_size)


class FrameSharder:
    """One rank of a frame rendered over `world` GPUs (one process each) through the C ABI.

    comm: an api.Comm (the library's RCCL gather), or None to gather with torch.distributed (`group`).
    Buffers are kept between frames."""

    def __init__(self, scene, rank: int, world: int, torch_device, comm=None, group=None, tile_size: int = TILE_SIZE):
        self.scene, self.rank, self.world, self.device = scene, rank, world, torch_device
        self.comm, self.group, self.tile_size = comm, group, tile_size
        self._tiles = None
        self._frame = None

    def render(self, cam, settings, stream=None):
        """Returns (frame, ray statistics of this rank).  frame: on rank 0 the (H, W, 3) float32 device tensor (native
        gather) or numpy array (torch gather); None elsewhere.  The gather is enqueued on `stream` (the current torch
        stream when None); the caller synchronises."""
        import torch

        from . import api

        stream = stream if stream is not None else torch.cuda.current_stream()
        n = api.tile_buffer_floats(settings, self.tile_size, self.rank, self.world)
        if self._tiles is None or self._tiles.numel() < max(n, 1):
            self._tiles = torch.empty(max(n, 1), dtype=torch.float32, device=self.device)
        stats = api.Renderer().render_tiles_device(self.scene, cam, settings, self.tile_size, self.rank, self.world, self._tiles, stream)
        if self.comm is None:
            return gather_compact(self._tiles[:n], settings, self.tile_size, self.rank, self.world, self.group), stats
        frame = None
        if self.rank == 0:
            shape = (settings.height, settings.width, 3)
            if self._frame is None or tuple(self._frame.shape) != shape:
                self._frame = torch.zeros(shape, dtype=torch.float32, device=self.device)
            frame = self._frame
        self.comm.gather_tiles_device(settings, self.tile_size, self._tiles, frame, stream)
        return frame, stats
