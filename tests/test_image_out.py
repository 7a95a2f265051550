"""Frame-buffer writers of the library (host code): OpenEXR as write_to_exr_file lays it out
(raytracer/src/lib/io/EXRWriter.cpp:6-72) and PNG with the tone mapping of write_to_png_file (io/PNG.cpp:7-28),
read back with a minimal reader each."""
import struct
import zlib

import numpy as np
import pytest

from raytracer_amd import api


def _read_exr(path):
    data = open(path, "rb").read()
    assert data[:8] == bytes([0x76, 0x2F, 0x31, 0x01, 2, 0, 0, 0])
    pos, attrs = 8, {}
    while data[pos] != 0:
        end = data.index(b"\0", pos); name = data[pos:end].decode(); pos = end + 1
        end = data.index(b"\0", pos); typ = data[pos:end].decode(); pos = end + 1
        size, = struct.unpack_from("<i", data, pos); pos += 4
        attrs[name] = (typ, data[pos:pos + size]); pos += size
    pos += 1
    x0, y0, x1, y1 = struct.unpack("<4i", attrs["dataWindow"][1])
    w, h = x1 - x0 + 1, y1 - y0 + 1
    chans, body, p = [], attrs["channels"][1], 0
    while body[p] != 0:
        end = body.index(b"\0", p); nm = body[p:end].decode(); p = end + 1
        ptype, = struct.unpack_from("<i", body, p); p += 16
        chans.append((nm, ptype))
    offsets = struct.unpack_from("<%dQ" % h, data, pos)
    img = {nm: np.empty((h, w), np.float32) for nm, _ in chans}
    for y in range(h):
        o = offsets[y]
        yy, nbytes = struct.unpack_from("<ii", data, o); o += 8
        assert yy == y
        for nm, ptype in chans:
            dt = np.float16 if ptype == 1 else np.float32
            img[nm][y] = np.frombuffer(data, dtype=dt, count=w, offset=o).astype(np.float32)
            o += w * np.dtype(dt).itemsize
    return attrs, chans, img


@pytest.mark.parametrize("full", [False, True])
def test_exr_channels_and_values(tmp_path, full):
    rng = np.random.RandomState(4)
    frame = (rng.rand(7, 11, 3).astype(np.float32) * 8.0) ** 3
    frame[0, 0] = (0.0, 1e-7, 70000.0)        # zero, a half subnormal, overflow to infinity
    frame[1, 2] = (1.0 + 2.0 ** -11, 1.0 + 3 * 2.0 ** -11, 65504.0)   # ties to even, largest half
    path = str(tmp_path / "f.exr")
    api.write_exr(path, frame, full_floats=full)
    attrs, chans, img = _read_exr(path)
    assert [c[0] for c in chans] == ["B", "G", "R"] and all(c[1] == (2 if full else 1) for c in chans)   # io/EXRWriter.cpp:46-57
    assert attrs["compression"][1] == b"\0" and attrs["lineOrder"][1] == b"\0"
    got = np.stack([img["R"], img["G"], img["B"]], axis=2)
    with np.errstate(over="ignore"):
        want = frame if full else frame.astype(np.float16).astype(np.float32)
    assert np.array_equal(got, want)


def test_png_tone_mapping(tmp_path):
    rng = np.random.RandomState(5)
    frame = rng.rand(9, 5, 3).astype(np.float32) * 12.0
    path = str(tmp_path / "f.png")
    api.write_png(path, frame)                # exposure -2.5, gamma 2.2: the executable's defaults
    data = open(path, "rb").read()
    assert data[:8] == b"\x89PNG\r\n\x1a\n"
    pos, idat, ihdr = 8, b"", None
    while pos < len(data):
        n, = struct.unpack_from(">I", data, pos)
        tag, body = data[pos + 4:pos + 8], data[pos + 8:pos + 8 + n]
        assert struct.unpack_from(">I", data, pos + 8 + n)[0] == zlib.crc32(tag + body) & 0xFFFFFFFF
        if tag == b"IHDR": ihdr = struct.unpack(">IIBBBBB", body)
        if tag == b"IDAT": idat += body
        pos += 12 + n
    assert ihdr == (5, 9, 8, 2, 0, 0, 0)
    raw = np.frombuffer(zlib.decompress(idat), np.uint8).reshape(9, 1 + 5 * 3)
    assert (raw[:, 0] == 0).all()
    got = raw[:, 1:].reshape(9, 5, 3)
    scaled = (frame.astype(np.float64) * 2.0 ** -2.5).astype(np.float32)
    toned = (scaled.astype(np.float64) ** (1 / 2.2)).astype(np.float32)
    want = np.minimum(255, (toned * np.float32(255)).astype(np.int32))
    assert np.abs(got.astype(int) - want).max() <= 1 and (got == want).mean() > 0.99   # libm pow may differ in the last bit
    with pytest.raises(api.RtgpuError, match="cannot write"):
        api.write_png(str(tmp_path / "no" / "dir.png"), frame)
