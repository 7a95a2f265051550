"""The TILE exchange format (raytracer/src/lib/io/TileFile.cpp:8-70): byte layout, round trip, the
reference's error cases, and the merge of band files written by the ranks of a sharded render."""
import struct

import numpy as np
import pytest

from raytracer_amd import distributed, tilefile


def frame(h=12, w=16, seed=0):
    return np.random.RandomState(seed).rand(h, w, 3).astype(np.float32)


def test_byte_layout_is_the_reference_layout(tmp_path):
    f = frame()
    p = str(tmp_path / "a.tile")
    tilefile.write_tile_file(f, (3, 2, 9, 7), p)
    raw = open(p, "rb").read()
    assert raw[:4] == b"TILE"
    assert struct.unpack("<6i", raw[4:28]) == (16, 12, 3, 2, 9, 7)      # W, H, then Tile{xStart, yStart, xEnd, yEnd}
    assert len(raw) == 28 + (7 - 2) * (9 - 3) * 12                       # RGB = 3 floats, no padding
    px = np.frombuffer(raw[28:], dtype="<f4").reshape(5, 6, 3)
    assert np.array_equal(px, f[2:7, 3:9])                               # rows of the tile, x fastest
    w, h, t, got = tilefile.read_tile_file(p)
    assert (w, h, t) == (16, 12, (3, 2, 9, 7)) and np.array_equal(got, f[2:7, 3:9])


def test_empty_tile_and_full_frame(tmp_path):
    f = frame()
    p = str(tmp_path / "e.tile")
    tilefile.write_tile_file(f, (4, 4, 4, 9), p)
    assert tilefile.read_tile_file(p)[3].shape == (5, 0, 3)
    tilefile.write_tile_file(f, (0, 0, 16, 12), p)
    out = np.zeros_like(f)
    assert tilefile.load_into_frame(out, p) == (0, 0, 16, 12) and np.array_equal(out, f)


def test_error_cases_of_the_reference(tmp_path):
    f = frame()
    p = str(tmp_path / "x.tile")
    open(p, "wb").write(b"TILF" + b"\0" * 24)
    with pytest.raises(RuntimeError, match="Bad file"):
        tilefile.read_tile_file(p)
    tilefile.write_tile_file(f, (0, 0, 4, 4), p)
    with pytest.raises(RuntimeError, match="Resolution mismatch"):
        tilefile.load_into_frame(np.zeros((12, 17, 3), np.float32), p)
    open(p, "wb").write(struct.pack("<4s6i", b"TILE", 16, 12, 5, 0, 4, 4))
    with pytest.raises(RuntimeError, match="Invalid tile geometry"):
        tilefile.read_tile_file(p)
    open(p, "wb").write(struct.pack("<4s6i", b"TILE", 16, 12, 0, 0, 17, 4))
    with pytest.raises(RuntimeError, match="Invalid tile geometry"):
        tilefile.read_tile_file(p)
    with pytest.raises(RuntimeError, match="Invalid tile geometry"):
        tilefile.write_tile_file(f, (0, 0, 4, 13), p)


def row_bands(height, world, rank, rows=4):
    """Rectangles of full width, `rows` high, band b for process b % world: how separate runs of the reference's executable
    would share a frame out with --ystart / --yend."""
    return [(y0, min(height, y0 + rows)) for b, y0 in enumerate(range(0, height, rows)) if b % world == rank]


def test_merge_of_rank_bands_rebuilds_the_frame(tmp_path):
    f = frame(h=37, w=20, seed=3)
    seen = np.zeros(37, dtype=int)
    for rank in range(3):
        for y0, y1 in row_bands(37, 3, rank):
            seen[y0:y1] += 1
    assert np.all(seen == 1)
    paths = []
    for rank in range(3):
        for k, (y0, y1) in enumerate(row_bands(37, 3, rank)):
            p = str(tmp_path / f"r{rank}_{k}.tile")
            tilefile.write_tile_file(f, (0, y0, 20, y1), p)
            paths.append(p)
    assert np.array_equal(tilefile.merge_tile_files(paths), f)
    # rows no file covers stay zero
    y1_first, y1_second = row_bands(37, 3, 0)[0][1], row_bands(37, 3, 1)[0][1]
    assert not tilefile.merge_tile_files(paths[:1])[y1_first:y1_second].any()
