"""The reference's TILE exchange format (raytracer/src/lib/io/TileFile.cpp:8-70).

The reference scales over processes by rendering disjoint tiles (`--xstart/--xend/--ystart/--yend
--outputtype tile`, exe/Raytracer.cpp:65-68) and merging the files (`--mergetiles`,
exe/Raytracer.cpp:260-277).  Writing this format lets a frame rendered here take part in such a
merge, and reading it lets tiles rendered by the reference be pasted into a frame rendered here.

Layout (native little-endian, no padding): "TILE", int32 frame width, int32 frame height, the Tile
object = int32 xStart, yStart, xEnd, yEnd (film/Tile.h:19-23), then the tile's pixels row by row,
three float32 (r, g, b) each (film/RGB.h:152-155).
"""
from __future__ import annotations

import struct
from typing import Iterable, Tuple

import numpy as np

MAGIC = b"TILE"
_HEADER = struct.Struct("<4s6i")


def write_tile_file(frame: np.ndarray, tile: Tuple[int, int, int, int], path: str) -> None:
    """write_to_tile_file: `frame` is the whole (height, width, 3) float32 frame buffer, `tile` the
    half-open (x_start, y_start, x_end, y_end) region that is stored."""
    frame = np.asarray(frame)
    if frame.ndim != 3 or frame.shape[2] != 3:
        raise ValueError("frame must have shape (height, width, 3)")
    h, w = frame.shape[:2]
    xs, ys, xe, ye = (int(v) for v in tile)
    _check_tile(xs, ys, xe, ye, w, h)
    with open(path, "wb") as f:
        f.write(_HEADER.pack(MAGIC, w, h, xs, ys, xe, ye))
        f.write(np.ascontiguousarray(frame[ys:ye, xs:xe], dtype="<f4").tobytes())


def _check_tile(xs, ys, xe, ye, w, h):
    # the checks of load_from_tile_file, io/TileFile.cpp:53-58
    if xs > xe or ys > ye or xs < 0 or xe < 0 or ys < 0 or ye < 0 or xe > w or ye > h:
        raise RuntimeError("Invalid tile geometry specification")


def read_tile_file(path: str):
    """Returns (frame width, frame height, (x_start, y_start, x_end, y_end), pixels of shape
    (y_end - y_start, x_end - x_start, 3))."""
    with open(path, "rb") as f:
        head = f.read(_HEADER.size)
        if len(head) < _HEADER.size or head[:4] != MAGIC:
            raise RuntimeError("Bad file")                     # io/TileFile.cpp:37-40
        _, w, h, xs, ys, xe, ye = _HEADER.unpack(head)
        _check_tile(xs, ys, xe, ye, w, h)
        n = (ye - ys) * (xe - xs) * 3
        data = np.frombuffer(f.read(n * 4), dtype="<f4")
        if data.size != n:
            raise RuntimeError("Bad file")
    return w, h, (xs, ys, xe, ye), data.reshape(ye - ys, xe - xs, 3).astype(np.float32)


def load_into_frame(frame: np.ndarray, path: str) -> Tuple[int, int, int, int]:
    """load_from_tile_file: pastes the file's tile into `frame` ((height, width, 3) float32)."""
    w, h, (xs, ys, xe, ye), px = read_tile_file(path)
    if frame.shape[1] != w or frame.shape[0] != h:
        raise RuntimeError("Resolution mismatch")              # io/TileFile.cpp:46-49
    frame[ys:ye, xs:xe] = px
    return xs, ys, xe, ye


def merge_tile_files(paths: Iterable[str]) -> np.ndarray:
    """--mergetiles (exe/Raytracer.cpp:260-277): a frame of the first file's resolution, black where
    no tile covers it, later files overwrite earlier ones."""
    paths = list(paths)
    if not paths:
        raise ValueError("no tile files")
    w, h, _, _ = read_tile_file(paths[0])
    frame = np.zeros((h, w, 3), dtype=np.float32)
    for p in paths:
        load_into_frame(frame, p)
    return frame
