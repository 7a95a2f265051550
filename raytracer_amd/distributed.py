"""Image-space sharding of a render over the ranks of one node and the frame-buffer gather.

The reference scales out with separate processes that render disjoint tiles
(--xstart/--xend/--ystart/--yend, exe/Raytracer.cpp:65-68), write TILE files (io/TileFile.cpp:8-27)
and paste them together with --mergetiles (exe/Raytracer.cpp:260-277, io/TileFile.cpp:29-70).  Here
one process per GPU renders an interleaved set of row bands of the frame (scene replicated on every
GPU) and the bands are gathered to rank 0 with one collective (RCCL over xGMI when the process
group is "nccl"; "gloo" on CPU tensors in the tests).  Pixels are keyed by their global pixel
index in the RNG, so the assembled image is bit-identical for every world size.
"""
from __future__ import annotations

from typing import Callable, List, Tuple

import numpy as np


def band_height(height: int, world: int, bands_per_rank: int = 4) -> int:
    """Rows per band: about bands_per_rank interleaved bands for every rank, at least 1 row."""
    return max(1, -(-height // (world * bands_per_rank)))


def rank_bands(height: int, world: int, rank: int, bands_per_rank: int = 4) -> List[Tuple[int, int]]:
    """Half-open row ranges [y0, y1) of `rank`: band b goes to rank b % world (interleaved so that
    an unevenly expensive image still balances)."""
    bh = band_height(height, world, bands_per_rank)
    out = []
    b = 0
    for y0 in range(0, height, bh):
        if b % world == rank:
            out.append((y0, min(height, y0 + bh)))
        b += 1
    return out


def padded_rows(height: int, world: int, bands_per_rank: int = 4) -> int:
    """Rows of the equal-sized per-rank message (the largest rank's row count)."""
    return max(sum(y1 - y0 for y0, y1 in rank_bands(height, world, r, bands_per_rank)) for r in range(world))


def render_sharded(render_band: Callable, width: int, height: int, rank: int, world: int, torch_device,
                   group=None, bands_per_rank: int = 4):
    """Renders this rank's bands and gathers the frame on rank 0.

    render_band(y0, y1, out) must fill `out` (a (y1-y0, width, 3) float32 torch tensor on
    torch_device) with the rows [y0, y1) of the image.  Returns the (height, width, 3) float32
    frame on rank 0 and None elsewhere.
    """
    import torch
    import torch.distributed as dist

    bands = rank_bands(height, world, rank, bands_per_rank)
    rows = padded_rows(height, world, bands_per_rank)
    local = torch.zeros((rows, width, 3), dtype=torch.float32, device=torch_device)
    r = 0
    for y0, y1 in bands:
        render_band(y0, y1, local[r:r + (y1 - y0)])
        r += y1 - y0
    if world == 1:
        gathered = [local]
    elif local.is_cuda and dist.get_backend(group) == "gloo":
        # a CPU process group (rehearsals without RCCL): stage the exchange through host memory
        host = local.cpu()
        parts = [torch.empty_like(host) for _ in range(world)] if rank == 0 else None
        dist.gather(host, parts, dst=0, group=group)
        gathered = [t.to(local.device) for t in parts] if rank == 0 else None
    else:
        gathered = [torch.empty_like(local) for _ in range(world)] if rank == 0 else None
        dist.gather(local, gathered, dst=0, group=group)
    if rank != 0:
        return None
    frame = torch.empty((height, width, 3), dtype=torch.float32, device=torch_device)
    for src in range(world):
        r = 0
        for y0, y1 in rank_bands(height, world, src, bands_per_rank):
            frame[y0:y1] = gathered[src][r:r + (y1 - y0)]
            r += y1 - y0
    return frame


def render_frame_distributed(scene, cam, settings, rank: int, world: int, torch_device, group=None,
                             bands_per_rank: int = 4, stream=None):
    """Sharded Renderer::render of the whole frame through the C ABI (GPU ranks).

    Returns (frame on rank 0 or None, ray statistics of this rank as a dict)."""
    import torch

    from . import abi, api

    W, H = settings.width, settings.height
    full = torch.zeros((H, W, 3), dtype=torch.float32, device=torch_device)
    totals = {"primary_rays": 0, "extension_rays": 0, "shadow_rays": 0, "device_ms": 0.0}
    renderer = api.Renderer()

    def render_band(y0, y1, out):
        st = abi.RenderSettings.from_buffer_copy(settings)
        st.x_start, st.x_end, st.y_start, st.y_end = 0, W, y0, y1
        stats = renderer.render_device(scene, cam, st, full, stream)
        out.copy_(full[y0:y1])
        totals["primary_rays"] += int(stats.primary_rays)
        totals["extension_rays"] += int(stats.extension_rays)
        totals["shadow_rays"] += int(stats.shadow_rays)
        totals["device_ms"] += float(stats.device_ms)

    frame = render_sharded(render_band, W, H, rank, world, torch_device, group, bands_per_rank)
    return frame, totals


def assemble_reference(height: int, width: int, world: int, fill: Callable[[int, int], np.ndarray], bands_per_rank: int = 4):
    """Host-side model of the gather: what rank 0 must end up with (used by the tests)."""
    frame = np.empty((height, width, 3), dtype=np.float32)
    for rnk in range(world):
        for y0, y1 in rank_bands(height, world, rnk, bands_per_rank):
            frame[y0:y1] = fill(y0, y1)
    return frame
