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


# ---- every bit depth, Adam7 interlacing, BMP -------------------------------------------------------------------------
ADAM7 = ((0, 0, 8, 8), (4, 0, 8, 8), (0, 4, 4, 8), (2, 0, 4, 4), (0, 2, 2, 4), (1, 0, 2, 2), (0, 1, 1, 2))


def pack_rows(smp, depth, filters):
    """smp: (h, w, c) integer samples at file depth -> filtered PNG scanlines (samples packed MSB first / big-endian)."""
    h, w, c = smp.shape
    if h == 0 or w == 0:
        return b""
    if depth == 16:
        rows = np.stack([smp >> 8, smp & 0xFF], axis=-1).reshape(h, w * c * 2).astype(np.uint8)
        fd = 2 * c
    elif depth == 8:
        rows = smp.reshape(h, w * c).astype(np.uint8)
        fd = c
    else:
        bits = np.zeros((h, ((w * c * depth + 7) // 8) * 8), np.uint8)
        flat = smp.reshape(h, w * c)
        for b in range(depth):
            bits[:, b:w * c * depth:depth] = (flat >> (depth - 1 - b)) & 1
        rows = np.packbits(bits, axis=1)
        fd = 1
    return filtered_bytes(rows, fd, filters)


def filtered_bytes(rows, fd, filters):
    h, n = rows.shape
    out = bytearray()
    r = rows.astype(np.int32)
    for y in range(h):
        ft = filters[y % len(filters)]
        cur, up = r[y], r[y - 1] if y else np.zeros(n, np.int32)
        line = bytearray([ft])
        for i in range(n):
            a = cur[i - fd] if i >= fd else 0
            b = up[i]
            cc = up[i - fd] if i >= fd else 0
            pred = [0, a, b, (a + b) >> 1, paeth(int(a), int(b), int(cc))][ft]
            line.append((int(cur[i]) - int(pred)) & 0xFF)
        out += line
    return bytes(out)


def make_png_any(smp, ctype, depth, interlace=False, filters=(0, 1, 2, 3, 4), extra=b""):
    h, w, _ = smp.shape
    if interlace:
        data = b"".join(pack_rows(smp[y0::dy, x0::dx], depth, filters) for x0, y0, dx, dy in ADAM7)
    else:
        data = pack_rows(smp, depth, filters)
    return (b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, depth, ctype, 0, 0, 1 if interlace else 0)) + extra +
            chunk(b"IDAT", zlib.compress(data, 6)) + chunk(b"IEND", b""))


@pytest.mark.parametrize("interlace", [False, True])
@pytest.mark.parametrize("ctype,c,depth", [(0, 1, 1), (0, 1, 2), (0, 1, 4), (0, 1, 8), (0, 1, 16), (2, 3, 8), (2, 3, 16), (4, 2, 8), (4, 2, 16),
                                           (6, 4, 8), (6, 4, 16)])
def test_png_every_depth_and_interlace(probe, tmp_path, ctype, c, depth, interlace):
    rng = np.random.default_rng(ctype * 100 + depth + (1000 if interlace else 0))
    for h, w in ((1, 1), (2, 3), (5, 9), (8, 8), (19, 23)):
        smp = rng.integers(0, 1 << depth, size=(h, w, c)).astype(np.int64)
        p = tmp_path / "d.png"
        p.write_bytes(make_png_any(smp, ctype, depth, interlace))
        want = (smp >> 8 if depth == 16 else smp * {1: 255, 2: 85, 4: 17, 8: 1}[depth]).astype(np.uint8)  # what stbi_load's 8-bit interface returns
        assert np.array_equal(load(probe, p), want), (h, w)


@pytest.mark.parametrize("interlace", [False, True])
@pytest.mark.parametrize("depth", [1, 2, 4, 8])
def test_png_palette_depths(probe, tmp_path, depth, interlace):
    rng = np.random.default_rng(depth + (50 if interlace else 0))
    n = 1 << depth
    pal = rng.integers(0, 256, size=(min(n, 200), 3), dtype=np.uint8)
    idx = rng.integers(0, len(pal), size=(13, 21, 1)).astype(np.int64)
    p = tmp_path / "p.png"
    p.write_bytes(make_png_any(idx, 3, depth, interlace, extra=chunk(b"PLTE", pal.tobytes())))
    assert np.array_equal(load(probe, p), pal[idx[..., 0]])


def test_png_colour_key_at_16_bits_and_low_depth(probe, tmp_path):
    rng = np.random.default_rng(9)
    smp = rng.integers(0, 4, size=(7, 9, 3)).astype(np.int64) * 0x4001  # few distinct 16-bit values
    key = smp[3, 4]
    p = tmp_path / "k.png"
    p.write_bytes(make_png_any(smp, 2, 16, extra=chunk(b"tRNS", struct.pack(">HHH", *map(int, key)))))
    got = load(probe, p)
    assert got.shape == (7, 9, 4) and np.array_equal(got[..., :3], (smp >> 8).astype(np.uint8))
    assert np.array_equal(got[..., 3] == 0, (smp == key).all(axis=2))
    g = rng.integers(0, 4, size=(6, 10, 1)).astype(np.int64)
    p.write_bytes(make_png_any(g, 0, 2, interlace=True, extra=chunk(b"tRNS", struct.pack(">H", 2))))
    got = load(probe, p)
    assert got.shape == (6, 10, 2) and np.array_equal(got[..., 0], (g[..., 0] * 85).astype(np.uint8))
    assert np.array_equal(got[..., 1] == 0, g[..., 0] == 2)


def make_bmp(img, bits, top_down=False, v4_alpha=False, pal=None):
    h, w, c = img.shape
    stride = ((w * bits + 31) // 32) * 4
    rows = bytearray()
    for y in (range(h) if top_down else range(h - 1, -1, -1)):
        if bits == 8:
            line = bytes(img[y, :, 0])
        elif bits == 24:
            line = img[y, :, ::-1].tobytes()  # BGR
        else:
            a = img[y, :, 3:4] if c == 4 else np.zeros((w, 1), np.uint8)
            line = np.concatenate([img[y, :, 2::-1][:, :3], a], axis=1).astype(np.uint8).tobytes()  # BGRA
        rows += line + bytes(stride - len(line))
    hsz = 108 if v4_alpha else 40
    palette = b"" if pal is None else b"".join(bytes([int(q[2]), int(q[1]), int(q[0]), 0]) for q in pal)
    head = struct.pack("<IiiHHIIiiII", hsz, w, -h if top_down else h, 1, bits, 3 if v4_alpha else 0, len(rows), 2835, 2835, len(pal) if pal is not None else 0, 0)
    if v4_alpha:
        head += struct.pack("<IIII", 0x00FF0000, 0x0000FF00, 0x000000FF, 0xFF000000) + bytes(108 - 40 - 16)
    off = 14 + len(head) + len(palette)
    return b"BM" + struct.pack("<IHHI", off + len(rows), 0, 0, off) + head + palette + bytes(rows)


def test_bmp_variants(probe, tmp_path):
    rng = np.random.default_rng(11)
    img = rng.integers(0, 256, size=(7, 5, 3), dtype=np.uint8)  # 5 pixels x 3 bytes: padded rows
    p = tmp_path / "a.bmp"
    for top_down in (False, True):
        p.write_bytes(make_bmp(img, 24, top_down))
        assert np.array_equal(load(probe, p), img)
    p.write_bytes(make_bmp(img, 32))  # 32 bits without an alpha mask: three channels, like stb
    assert np.array_equal(load(probe, p), img)
    rgba = rng.integers(0, 256, size=(4, 6, 4), dtype=np.uint8)
    p.write_bytes(make_bmp(rgba, 32, v4_alpha=True))
    assert np.array_equal(load(probe, p), rgba)
    pal = rng.integers(0, 256, size=(17, 3), dtype=np.uint8)
    idx = rng.integers(0, 17, size=(9, 7, 1), dtype=np.uint8)
    p.write_bytes(make_bmp(idx, 8, pal=pal))
    assert np.array_equal(load(probe, p), pal[idx[..., 0]])
    bad = bytearray(make_bmp(img, 24))
    bad[30] = 1  # RLE8: refused, never misread
    p.write_bytes(bytes(bad))
    assert subprocess.run([probe, str(p)], capture_output=True).returncode == 1
    p.write_bytes(make_bmp(img, 24)[:60])
    assert subprocess.run([probe, str(p)], capture_output=True).returncode == 1


# ---- the PNG writer (llcompd's output): adaptive filters + own deflate, read back by zlib ------------------------------
def decode_png_py(data):
    """minimal PNG decoder (8-bit, non-interlaced) on Python's zlib: what any other PNG reader would see"""
    assert data[:8] == b"\x89PNG\r\n\x1a\n"
    o, idat, hdr = 8, b"", None
    while o < len(data):
        n, t = struct.unpack(">I4s", data[o:o + 8])
        body = data[o + 8:o + 8 + n]
        assert struct.unpack(">I", data[o + 8 + n:o + 12 + n])[0] == zlib.crc32(t + body) & 0xFFFFFFFF, "chunk CRC"
        if t == b"IHDR":
            hdr = struct.unpack(">IIBBBBB", body)
        elif t == b"IDAT":
            idat += body
        o += 12 + n
    w, h, depth, ctype, _, _, inter = hdr
    assert depth == 8 and inter == 0
    c = {0: 1, 2: 3, 4: 2, 6: 4}[ctype]
    raw = zlib.decompress(idat)
    row = w * c
    assert len(raw) == (row + 1) * h
    img = np.zeros((h, row), np.int32)
    for y in range(h):
        ft = raw[(row + 1) * y]
        line = np.frombuffer(raw, np.uint8, row, (row + 1) * y + 1).astype(np.int32)
        up = img[y - 1] if y else np.zeros(row, np.int32)
        for i in range(row):
            a = img[y, i - c] if i >= c else 0
            b = up[i]
            cc = up[i - c] if i >= c else 0
            pred = [0, a, b, (a + b) >> 1, paeth(int(a), int(b), int(cc))][ft]
            img[y, i] = (line[i] + pred) & 0xFF
    return img.astype(np.uint8).reshape(h, w, c), len(raw), len(idat)


@pytest.mark.parametrize("c,ctype", [(1, 0), (2, 4), (3, 2), (4, 6)])
def test_png_writer_round_trip_and_compression(probe, tmp_path, c, ctype):
    rng = np.random.default_rng(100 + c)
    h, w = 41, 67
    y, x, k = np.meshgrid(np.arange(h), np.arange(w), np.arange(c), indexing="ij")
    smooth = ((x * 2 + y * 3 + k * 50) & 0xFF).astype(np.uint8)                # the filters and LZ77 have something to find
    noisy = rng.integers(0, 256, size=(h, w, c), dtype=np.uint8)              # incompressible: must still be a valid stream
    flat = np.full((h, w, c), 77, np.uint8)                                    # maximal matches (length 258)
    for name, img in (("smooth", smooth), ("noisy", noisy), ("flat", flat), ("tiny", noisy[:1, :1])):
        src, dst = tmp_path / "in.png", tmp_path / "out.png"
        src.write_bytes(make_png(img, ctype))
        assert subprocess.run([probe, str(src), str(dst)]).returncode == 0
        got, raw_len, z_len = decode_png_py(dst.read_bytes())
        assert np.array_equal(got, img), name
        assert np.array_equal(load(probe, dst), img), name  # and our own reader agrees
        if name in ("smooth", "flat"):
            assert z_len < raw_len // 4, (name, z_len, raw_len)


def make_tga(img, rle=False, top_down=False, right_left=False):
    h, w, c = img.shape
    src = img if top_down else img[::-1]
    if right_left:
        src = src[:, ::-1]
    file_px = (src if c == 1 else np.concatenate([src[..., 2::-1][..., :3], src[..., 3:]], axis=2)).reshape(-1, c)  # B, G, R(, A)
    head = bytes([0, 0, (11 if c == 1 else 10) if rle else (3 if c == 1 else 2), 0, 0, 0, 0, 0, 0, 0, 0, 0]) + struct.pack("<HHBB", w, h, 8 * c,
                 (0x20 if top_down else 0) | (0x10 if right_left else 0) | (8 if c == 4 else 0))
    if not rle:
        return head + file_px.tobytes()
    out, i, n = bytearray(), 0, len(file_px)
    while i < n:  # runs where the next pixels repeat, raw packets otherwise (runs may cross scanlines: allowed by the readers)
        run = 1
        while i + run < n and run < 128 and (file_px[i + run] == file_px[i]).all():
            run += 1
        if run > 1:
            out += bytes([0x80 | (run - 1)]) + file_px[i].tobytes()
            i += run
        else:
            j = i + 1
            while j < n and j - i < 128 and not (j + 1 < n and (file_px[j + 1] == file_px[j]).all()):
                j += 1
            out += bytes([j - i - 1]) + file_px[i:j].tobytes()
            i = j
    return head + bytes(out)


@pytest.mark.parametrize("c", [1, 3, 4])
def test_tga_variants(probe, tmp_path, c):
    rng = np.random.default_rng(40 + c)
    img = rng.integers(0, 4, size=(13, 17, c), dtype=np.uint8) * 60  # few values: real runs for the RLE packets
    img[5] = 200
    p = tmp_path / "a.tga"
    for rle in (False, True):
        for top_down, right_left in ((False, False), (True, False), (False, True)):
            p.write_bytes(make_tga(img, rle, top_down, right_left))
            assert np.array_equal(load(probe, p), img), (rle, top_down, right_left)
    p.write_bytes(make_tga(img)[:40])
    assert subprocess.run([probe, str(p)], capture_output=True).returncode == 1
    bad = bytearray(make_tga(img))
    bad[2] = 1  # colour-mapped: refused
    p.write_bytes(bytes(bad))
    assert subprocess.run([probe, str(p)], capture_output=True).returncode == 1
