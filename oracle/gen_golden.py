#!/usr/bin/env python3
"""Generate tests/golden/*.json from the REAL reference (oracle/_ref/libllcomp_ref.so, i.e.
/root/reference/llcomp.hpp compiled in place by oracle/Makefile).  Run in the build container only:

    make -C oracle && python3 oracle/gen_golden.py

The fixtures are DATA (inputs are named generators, outputs are stream bytes / lengths / FNV-1a-64
hashes).  The plain-C restatement (liborc.so) is NOT used to produce them; it is what they pin.

 kat_streams.json     whole-image legacy streams: O1 (unmodified compressImage) where it is defined,
                      otherwise O2 (reference components + growable sink), with O1==O2 recorded.
 slice_payloads.json  parity level P1: per-slice payloads = reference stream of the cropped sub-image
                      (interleaved) or of the cropped post-RCT int16 plane (planar), plus the sliced
                      container assembled here in Python from those reference payloads.
 decode_behaviour.json  what the unmodified decompressImage does with damaged / random streams
                      (error class or decoded-pixel hash).
"""
import json
import os
import struct
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from orc import GENERATORS, Ref, fnv1a64  # noqa: E402

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden")
HEX_LIMIT = 4096


def fnv(b):
    if len(b) < 1 << 16:
        return "%016x" % fnv1a64(b)
    # vectorised-enough for big streams: chunked python loop is too slow, use the C helper in liborc
    # ONLY as a hash function (it does not touch codec code).
    import ctypes as C

    L = C.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "liborc.so"))
    L.orc_fnv1a64.restype = C.c_uint64
    L.orc_fnv1a64.argtypes = [C.c_char_p, C.c_size_t]
    return "%016x" % L.orc_fnv1a64(b, len(b))


def special(name, w, h, c):
    if name.startswith("g3@"):  # std::mt19937 noise with another seed
        return GENERATORS["g3"](w, h, c, seed=int(name[3:]))
    if name == "const0":
        return np.zeros((h, w, c), np.uint8)
    if name == "const255":
        return np.full((h, w, c), 255, np.uint8)
    return GENERATORS[name](w, h, c)


def kat_streams(ref):
    cases = [("g1", w, h, c) for (w, h, c) in [(1, 1, 3), (2, 2, 3), (5, 1, 3), (1, 5, 3), (4, 4, 3), (8, 8, 3), (4, 4, 4), (8, 2, 1), (16, 16, 3), (64, 64, 3),
                                                (1, 1, 1), (1, 1, 2), (1, 1, 4), (3, 2, 2), (7, 5, 2), (9, 1, 4), (1, 9, 4), (17, 3, 4), (2, 1, 3), (1, 2, 3), (33, 31, 5)]]
    cases += [("g2", 256, 256, 1), ("g2", 256, 256, 3), ("g2", 1920, 1080, 3), ("g2", 3840, 2160, 3), ("g2", 8192, 8192, 3)]
    cases += [("g3", 512, 512, 3), ("g3", 64, 64, 3), ("g3", 64, 64, 1), ("g3", 31, 17, 4), ("g3", 1920, 1080, 3), ("g3", 3840, 2160, 3)]
    cases += [("mid", 640, 360, 3), ("mid", 3840, 2160, 3)]
    cases += [(n, w, h, c) for n in ("checker", "const0", "const255") for (w, h, c) in [(16, 16, 3), (64, 48, 3), (64, 48, 1), (13, 7, 4)]]
    out = []
    for name, w, h, c in cases:
        img = special(name, w, h, c)
        s2 = ref.o2_compress_image(img)
        s1 = ref.o1_compress_image(img, len(s2))
        rec = {"gen": name, "w": w, "h": h, "c": c, "len": len(s2), "fnv1a64": fnv(s2),
               "source": "O1" if s1 is not None else "O2", "o1_equals_o2": None if s1 is None else (s1 == s2)}
        if s1 is not None:
            assert s1 == s2, (name, w, h, c)
        if len(s2) <= HEX_LIMIT:
            rec["hex"] = s2.hex()
        if c >= 3 and img.size <= 3840 * 2160 * 3:
            rc, px = ref.o1_decompress_image(s2)
            rec["ref_decode_roundtrip"] = bool(rc == 0 and np.array_equal(px, img))
            assert rec["ref_decode_roundtrip"]
        out.append(rec)
        print("kat", name, w, h, c, len(s2), rec["source"], flush=True)
    return out


def container(w, h, c, tile_w, tile_h, planar, payloads):
    head = bytes([0x9C, 1, c, 1 if planar else 0]) + struct.pack("<5I", w, h, tile_w, tile_h, len(payloads))
    return head + b"".join(struct.pack("<I", len(p)) for p in payloads) + b"".join(payloads)


# round 4: slicings at the capacity classes of the 2-D encoder's snapshot pass (1024 / 2048 / 4096 samples per slice), the
# benchmarked slicing for the photo-like content, four interleaved channels on a saturated checkerboard, one channel, ragged
# tiles -- appended by `gen_golden.py slice_add` (the vectors above are not regenerated)
SLICE_CASES_R4 = [
    ("nat", 1920, 1080, 3, 128, 8, True), ("g3", 1920, 1080, 3, 32, 32, False), ("mid", 1920, 1080, 3, 480, 4, True),
    ("nat", 3840, 2160, 3, 480, 1, True), ("checker", 640, 480, 4, 33, 31, False), ("g3", 1000, 700, 1, 64, 64, True),
    ("nat", 1920, 1080, 3, 64, 32, True),
]


def slice_payloads(ref, cases=None):
    cases = cases or [
        ("g1", 16, 16, 3, 8, 8, False), ("g1", 16, 16, 3, 8, 8, True), ("g1", 19, 13, 3, 8, 4, False), ("g1", 19, 13, 3, 8, 4, True),
        ("g3", 19, 13, 4, 5, 5, True), ("g3", 19, 13, 1, 5, 5, False), ("g3", 19, 13, 2, 19, 1, True), ("g1", 40, 6, 3, 40, 1, False),
        ("g1", 40, 6, 3, 40, 1, True), ("mid", 70, 50, 3, 32, 32, True), ("mid", 70, 50, 3, 32, 32, False), ("checker", 20, 20, 3, 7, 7, True),
        ("g3", 256, 128, 3, 64, 64, True), ("g3", 256, 128, 3, 64, 64, False), ("g2", 1920, 1080, 3, 1920, 1, False), ("g3", 1920, 1080, 3, 1920, 1, False),
        ("g3", 1920, 1080, 3, 64, 64, True), ("mid", 1920, 1080, 3, 128, 128, True), ("g2", 3840, 2160, 3, 64, 64, True), ("g3", 3840, 2160, 3, 64, 64, True),
        ("mid", 3840, 2160, 3, 64, 64, True), ("g3", 3840, 2160, 3, 256, 256, False),
        # the benchmarked slicing (bench.py default: per-channel planes, 480x1) at full 4K size, and two frames of BASELINE
        # config 5's batch (frame i = std::mt19937(1234 + i)): frame 0 is "g3" itself, frame 63 = seed 1297
        ("g3", 3840, 2160, 3, 480, 1, True), ("g2", 3840, 2160, 3, 480, 1, True), ("mid", 3840, 2160, 3, 480, 1, True),
        ("nat", 3840, 2160, 3, 64, 64, True), ("g3@1297", 3840, 2160, 3, 480, 1, True),
    ]
    out = []
    for name, w, h, c, tw, th, planar in cases:
        img = special(name, w, h, c)
        planes = ref.o2_forward_rct(img)
        pays = []
        for y0 in range(0, h, th):
            for x0 in range(0, w, tw):
                crop_px = img[y0:y0 + th, x0:x0 + tw]
                crop_s = planes[y0:y0 + th, x0:x0 + tw]
                if planar:
                    for k in range(c):
                        pays.append(ref.o2_encode_samples(np.ascontiguousarray(crop_s[:, :, k:k + 1])))
                else:
                    # the reference stream of the cropped sub-image, minus its 6-byte header
                    s2 = ref.o2_compress_image(np.ascontiguousarray(crop_px))
                    s1 = ref.o1_compress_image(np.ascontiguousarray(crop_px), len(s2))
                    assert s1 is None or s1 == s2
                    pays.append(s2[6:])
        cont = container(w, h, c, tw, th, planar, pays)
        rec = {"gen": name, "w": w, "h": h, "c": c, "tile_w": tw, "tile_h": th, "planar": planar, "n_slices": len(pays),
               "container_len": len(cont), "container_fnv1a64": fnv(cont)}
        if len(pays) <= 64:
            rec["slices"] = [{"len": len(p), "fnv1a64": fnv(p), **({"hex": p.hex()} if len(p) <= 256 else {})} for p in pays]
        if len(cont) <= HEX_LIMIT:
            rec["container_hex"] = cont.hex()
        out.append(rec)
        print("slice", name, w, h, c, tw, th, planar, len(pays), len(cont), flush=True)
    return out


def c4_bench_slicing(ref):
    """BASELINE config 4 at the slicing bench.py and the sharded path use (8192 x 8192 RGB8 noise, per-channel planes,
    512x1): the container assembled from the real reference's per-slice streams, as length + hash (393 216 slices)."""
    name, w, h, c, tw, th = "g3", 8192, 8192, 3, 512, 1
    img = special(name, w, h, c)
    planes = ref.o2_forward_rct(img)
    pays = []
    for y0 in range(h):
        row = planes[y0]
        for x0 in range(0, w, tw):
            for k in range(c):
                pays.append(ref.o2_encode_samples(np.ascontiguousarray(row[None, x0:x0 + tw, k:k + 1])))
        if y0 % 512 == 511:
            print("c4 rows", y0 + 1, flush=True)
    cont = container(w, h, c, tw, th, True, pays)
    lens = np.array([len(p) for p in pays], dtype=np.uint32)
    return [{"gen": name, "w": w, "h": h, "c": c, "tile_w": tw, "tile_h": th, "planar": True, "n_slices": len(pays), "container_len": len(cont),
             "container_fnv1a64": fnv(cont), "payload_bytes": int(lens.sum()), "slice_table_fnv1a64": fnv(lens.astype("<u4").tobytes())}]


def small_model(ref_small):
    """The bitstream of a reference built with LargeModel = false (oracle/_ref/libllcomp_ref_small.so): legacy streams
    (O1 where defined, else O2) and sliced containers assembled from that reference's per-slice payloads."""
    assert ref_small.lib.ref_large_model() == 0 and ref_small.lib.ref_states_nb() == (11 * 11 * 11 + 1) // 2 * 8
    out = []
    for name, w, h, c in [("g1", 1, 1, 3), ("g1", 8, 8, 3), ("g1", 16, 16, 3), ("g1", 17, 3, 4), ("g1", 8, 2, 1), ("g3", 64, 64, 3), ("mid", 97, 41, 3),
                          ("checker", 16, 16, 3), ("g1", 33, 31, 5), ("mid", 640, 360, 3), ("g3", 1920, 1080, 3), ("g2", 1920, 1080, 3)]:
        img = special(name, w, h, c)
        s2 = ref_small.o2_compress_image(img)
        s1 = ref_small.o1_compress_image(img, len(s2))
        assert s1 is None or s1 == s2
        rec = {"kind": "legacy", "gen": name, "w": w, "h": h, "c": c, "len": len(s2), "fnv1a64": fnv(s2), "source": "O1" if s1 is not None else "O2"}
        if len(s2) <= HEX_LIMIT:
            rec["hex"] = s2.hex()
        if c >= 3:
            rc, px = ref_small.o1_decompress_image(s2)
            assert rc == 0 and np.array_equal(px, img)
        out.append(rec)
        print("small legacy", name, w, h, c, len(s2), flush=True)
    for name, w, h, c, tw, th, planar in [("mid", 70, 50, 3, 32, 32, True), ("g3", 70, 50, 4, 32, 16, False), ("g1", 40, 6, 3, 40, 1, True), ("g1", 40, 6, 3, 13, 1, False),
                                          ("mid", 1920, 1080, 3, 64, 64, True), ("g3", 1920, 1080, 3, 480, 1, True)]:
        img = special(name, w, h, c)
        planes = ref_small.o2_forward_rct(img)
        pays = []
        for y0 in range(0, h, th):
            for x0 in range(0, w, tw):
                if planar:
                    for k in range(c):
                        pays.append(ref_small.o2_encode_samples(np.ascontiguousarray(planes[y0:y0 + th, x0:x0 + tw, k:k + 1])))
                else:
                    pays.append(ref_small.o2_compress_image(np.ascontiguousarray(img[y0:y0 + th, x0:x0 + tw]))[6:])
        cont = bytearray(container(w, h, c, tw, th, planar, pays))
        cont[3] |= 2  # the container's small-model flag
        cont = bytes(cont)
        rec = {"kind": "sliced", "gen": name, "w": w, "h": h, "c": c, "tile_w": tw, "tile_h": th, "planar": planar, "n_slices": len(pays),
               "container_len": len(cont), "container_fnv1a64": fnv(cont)}
        if len(cont) <= HEX_LIMIT:
            rec["container_hex"] = cont.hex()
        out.append(rec)
        print("small sliced", name, w, h, c, tw, th, planar, len(cont), flush=True)
    return out


def craft_bins(ref, bins):
    """Range-code an arbitrary (slot, bit) list for ONE context with fresh state.  Only used to build
    damaged-stream test inputs that no legal image produces; probabilities and state steps come from
    the reference's own State class (ref_state_p / ref_state_next), the carry logic follows the usual
    held-byte scheme.  Whether the result means what was intended is decided by the real decoder."""
    state = [0] * 8
    low, rng, held, pend, out = 0, 0xFF00, -1, 0, bytearray()

    def renorm():
        nonlocal low, rng, held, pend
        while rng < 0x100:
            if held < 0:
                held = low >> 8
            elif low <= 0xFF00:
                out.append(held); out.extend(b"\xff" * pend); pend = 0; held = low >> 8
            elif low >= 0x10000:
                out.append((held + 1) & 0xFF); out.extend(b"\x00" * pend); pend = 0; held = (low >> 8) & 0xFF
            else:
                pend += 1
            low = (low & 0xFF) << 8
            rng <<= 8

    for slot, bit in bins:
        p = ref.lib.ref_state_p(state[slot])
        r1 = (rng * p) >> 8
        if bit:
            low += rng - r1; rng = r1
        else:
            rng -= r1
        state[slot] = ref.lib.ref_state_next(state[slot], bit)
        renorm()
    rng = 0xFF; low += 0xFF; renorm(); rng = 0xFF; renorm()
    return bytes(out)


def decode_behaviour(ref):
    """Unmodified decompressImage on damaged input (3-channel only: D2).  Streams are bytes from
    numpy's PCG64(seed) or edits of a valid stream, recorded in full (hex)."""
    out = []
    valid = ref.o2_compress_image(special("g1", 16, 16, 3))
    edits = {"valid": valid, "bad_magic_77": bytes([0x77]) + valid[1:], "bad_magic_9d": bytes([0x9D]) + valid[1:],
             "truncated_half": valid[: len(valid) // 2], "truncated_header_only": valid[:6], "truncated_7": valid[:7],
             "tail_flipped": valid[:-8] + bytes(b ^ 0xFF for b in valid[-8:]), "mid_flipped": valid[:40] + bytes(b ^ 0x55 for b in valid[40:48]) + valid[48:]}
    rng = np.random.Generator(np.random.PCG64(99))
    for i in range(24):
        body = rng.integers(0, 256, size=int(rng.integers(8, 200)), dtype=np.uint8).tobytes()
        w, h = int(rng.integers(1, 24)), int(rng.integers(1, 24))
        edits[f"random_{i}"] = bytes([0x79, 3, w, 0, h, 0]) + body
    for fill in (0x00, 0xFF, 0x80, 0x7F):
        edits[f"fill_{fill:02x}"] = bytes([0x79, 3, 12, 0, 9, 0]) + bytes([fill]) * 300
    # streams whose first residual has a unary exponent run of n ones (n > 31 makes the reference throw
    # "Invalid exponent", llcomp.hpp:230-235).  Crafted with craft_bins(), judged by the real decoder.
    for n in (30, 31, 32, 33, 40):
        bins = [(0, 0)] + [(min(1 + i, 4), 1) for i in range(n)] + [(min(1 + n, 4), 0)]
        edits[f"exponent_run_{n}"] = bytes([0x79, 3, 4, 0, 4, 0]) + craft_bins(ref, bins) + bytes(16)
    for name, s in edits.items():
        rc, px = ref.o1_decompress_image(s)
        rec = {"name": name, "hex": s.hex(), "rc": int(rc)}
        if rc == 0:
            rec["w"], rec["h"], rec["c"] = int(px.shape[1]), int(px.shape[0]), int(px.shape[2])
            rec["pixels_fnv1a64"] = fnv(px.tobytes())
        out.append(rec)
        print("decode", name, rc, flush=True)
    return out


def main():
    ref = Ref()
    os.makedirs(OUT, exist_ok=True)
    meta = {"made_by": "oracle/gen_golden.py", "reference": "vovach777/llcomp llcomp.hpp (magic 0x%02x, %d states)" % (ref.lib.ref_magic(), ref.lib.ref_states_nb()),
            "generators": "oracle/orc.py GENERATORS (g1,g2,g3=std::mt19937(1234)&255,mid,checker) + const0/const255",
            "hash": "FNV-1a-64 over the bytes"}
    only = set(sys.argv[1:])  # e.g. `gen_golden.py decode` regenerates one file
    if "slice_add" in only:  # append the round-4 slicings to slice_payloads.json, leaving what is there untouched
        path = os.path.join(OUT, "slice_payloads.json")
        doc = json.load(open(path))
        have = {(v["gen"], v["w"], v["h"], v["c"], v["tile_w"], v["tile_h"], v["planar"]) for v in doc["vectors"]}
        new = [cs for cs in SLICE_CASES_R4 if cs not in have]
        if new:
            doc["vectors"] += slice_payloads(ref, new)
            with open(path, "w") as f:
                json.dump(doc, f, indent=1)
        print("slice_payloads.json:", len(new), "vectors added")
        return
    for key, fn, make in (("kat", "kat_streams.json", kat_streams), ("slice", "slice_payloads.json", slice_payloads),
                          ("decode", "decode_behaviour.json", decode_behaviour)):
        if only and key not in only:
            continue
        with open(os.path.join(OUT, fn), "w") as f:
            json.dump({"meta": meta, "vectors": make(ref)}, f, indent=1)
        print("wrote", fn)
    if "c4" in only:  # (minutes of single-thread reference coding: only on request)
        with open(os.path.join(OUT, "c4_bench_slicing.json"), "w") as f:
            json.dump({"meta": meta, "vectors": c4_bench_slicing(ref)}, f, indent=1)
        print("wrote c4_bench_slicing.json")
    if not only or "small" in only:
        from orc import REF_SMALL_PATH

        m2 = dict(meta, reference=meta["reference"] + ", compiled with LargeModel = false (oracle/Makefile: _ref/libllcomp_ref_small.so)")
        with open(os.path.join(OUT, "small_model.json"), "w") as f:
            json.dump({"meta": m2, "vectors": small_model(Ref(REF_SMALL_PATH))}, f, indent=1)
        print("wrote small_model.json")
    if only:
        return
    # primitive tables straight from the reference's own functions
    prim = {"meta": meta,
            "quant11": [ref.lib.ref_quant11(x) for x in range(-600, 601)],
            "quant5": [ref.lib.ref_quant5(x) for x in range(-600, 601)],
            "quant_domain": [-600, 600],
            "state_p": [ref.lib.ref_state_p(s) for s in range(128)],
            "state_next0": [ref.lib.ref_state_next(s, 0) for s in range(128)],
            "state_next1": [ref.lib.ref_state_next(s, 1) for s in range(128)]}
    with open(os.path.join(OUT, "primitives.json"), "w") as f:
        json.dump(prim, f)
    print("wrote primitives.json")


if __name__ == "__main__":
    main()
