"""The CLIs' image reader (tools/image_io.hpp) stands in for stb_image, which the reference delegates file decoding to
(llcompc.cpp:25) and which is neither vendored nor installed.  CPU-only: PNGs made with Python's zlib in every deflate
block type, every row filter, every colour type the reader claims, plus PNM/PAM; the pixels must come out exactly, with
the channel counts stb would report."""
import os
import struct
import subprocess
import zlib

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def probe(tmp_path_factory):
    exe = str(tmp_path_factory.mktemp("probe") / "png_probe")
    subprocess.check_call(["g++", "-O1", "-std=c++17", "-o", exe, os.path.join(ROOT, "tests", "helpers", "png_probe.cpp")])
    return exe


def chunk(t, body):
    return struct.pack(">I", len(body)) + t + body + struct.pack(">I", zlib.crc32(t + body) & 0xFFFFFFFF)


def paeth(a, b, c):
    p = a + b - c
    pa, pb, pc = abs(p - a), abs(p - b), abs(p - c)
    return a if pa <= pb and pa <= pc else (b if pb <= pc else c)


def filtered_rows(img, filters):
    """img: (h, w, c) uint8 -> PNG scanlines with the given filter type per row (cycled)."""
    h, w, c = img.shape
    rows = img.reshape(h, w * c).astype(np.int32)
    out = bytearray()
    for y in range(h):
        ft = filters[y % len(filters)]
        cur, up = rows[y], rows[y - 1] if y else np.zeros(w * c, np.int32)
        line = bytearray([ft])
        for i in range(w * c):
            a = cur[i - c] if i >= c else 0
            b = up[i]
            cc = up[i - c] if i >= c else 0
            pred = [0, a, b, (a + b) >> 1, paeth(int(a), int(b), int(cc))][ft]
            line.append((int(cur[i]) - int(pred)) & 0xFF)
        out += line
    return bytes(out)


def make_png(img, ctype, filters=(0,), level=6, strategy=zlib.Z_DEFAULT_STRATEGY, extra=b"", split=1):
    h, w, _ = img.shape
    co = zlib.compressobj(level, zlib.DEFLATED, 15, 9, strategy)
    z = co.compress(filtered_rows(img, filters)) + co.flush()
    parts = [z[i * len(z) // split:(i + 1) * len(z) // split] for i in range(split)]
    return (b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, ctype, 0, 0, 0)) + extra +
            b"".join(chunk(b"IDAT", p) for p in parts) + chunk(b"IEND", b""))


def load(probe, path):
    r = subprocess.run([probe, str(path)], capture_output=True)
    assert r.returncode == 0, r.stderr.decode()
    head, _, body = r.stdout.partition(b"\n")
    w, h, c = map(int, head.split())
    return np.frombuffer(body, np.uint8).reshape(h, w, c)


@pytest.mark.parametrize("ctype,c", [(0, 1), (2, 3), (4, 2), (6, 4)])
@pytest.mark.parametrize("level,strategy", [(0, zlib.Z_DEFAULT_STRATEGY), (1, zlib.Z_FIXED), (9, zlib.Z_DEFAULT_STRATEGY)])
def test_png_colour_types_filters_and_block_types(probe, tmp_path, ctype, c, level, strategy):
    rng = np.random.default_rng(ctype * 10 + level)
    h, w = 37, 53
    y, x, k = np.meshgrid(np.arange(h), np.arange(w), np.arange(c), indexing="ij")
    img = ((x * 5 + y * 3 + k * 40 + rng.integers(0, 6, size=(h, w, c))) & 0xFF).astype(np.uint8)  # compressible: real matches
    p = tmp_path / "a.png"
    p.write_bytes(make_png(img, ctype, filters=(0, 1, 2, 3, 4), level=level, strategy=strategy, split=3))
    assert np.array_equal(load(probe, p), img)


def test_png_palette_and_transparency(probe, tmp_path):
    rng = np.random.default_rng(5)
    idx = rng.integers(0, 7, size=(20, 31, 1), dtype=np.uint8)
    pal = rng.integers(0, 256, size=(7, 3), dtype=np.uint8)
    p = tmp_path / "p.png"
    p.write_bytes(make_png(idx, 3, filters=(0, 2), extra=chunk(b"PLTE", pal.tobytes())))
    assert np.array_equal(load(probe, p), pal[idx[..., 0]])
    alpha = bytes([0, 128, 255, 7])  # shorter than the palette: the rest is opaque
    p.write_bytes(make_png(idx, 3, extra=chunk(b"PLTE", pal.tobytes()) + chunk(b"tRNS", alpha)))
    want = np.concatenate([pal[idx[..., 0]], np.array(list(alpha) + [255] * 3, np.uint8)[idx[..., 0]][..., None]], axis=2)
    assert np.array_equal(load(probe, p), want)
    # colour key on RGB: stb reports 4 channels, alpha 0 where the pixel equals the key
    img = rng.integers(0, 3, size=(9, 11, 3), dtype=np.uint8)
    key = img[4, 5]
    p.write_bytes(make_png(img, 2, extra=chunk(b"tRNS", struct.pack(">HHH", *map(int, key)))))
    got = load(probe, p)
    assert got.shape == (9, 11, 4) and np.array_equal(got[..., :3], img)
    assert np.array_equal(got[..., 3] == 0, (img == key).all(axis=2))


def test_pnm_and_rejections(probe, tmp_path):
    img = np.arange(5 * 4 * 3, dtype=np.uint8).reshape(5, 4, 3)
    (tmp_path / "a.ppm").write_bytes(b"P6\n# comment\n4 5\n255\n" + img.tobytes())
    assert np.array_equal(load(probe, tmp_path / "a.ppm"), img)
    (tmp_path / "a.pam").write_bytes(b"P7\nWIDTH 4\nHEIGHT 5\nDEPTH 3\nMAXVAL 255\nTUPLTYPE RGB\nENDHDR\n" + img.tobytes())
    assert np.array_equal(load(probe, tmp_path / "a.pam"), img)
    bad = make_png(img, 2)
    for name, data in (("trunc.png", bad[:60]), ("sig.png", b"\x89PNX" + bad[4:]), ("gif.gif", b"GIF89a" + bytes(50)),
                       ("depth16.png", bad[:24] + b"\x10" + bad[25:]), ("interlaced.png", bad[:28] + b"\x01" + bad[29:])):
        (tmp_path / name).write_bytes(data)
        assert subprocess.run([probe, str(tmp_path / name)], capture_output=True).returncode == 1, name
    z = bytearray(bad)
    z[len(z) // 2] ^= 0x55  # damaged deflate data: an error or some picture, never a crash
    (tmp_path / "noise.png").write_bytes(bytes(z))
    assert subprocess.run([probe, str(tmp_path / "noise.png")], capture_output=True).returncode in (0, 1)
