#!/usr/bin/env python3
"""Soak of round 6's new paths with seeds the committed tests do not use: slices above 4096 samples (chunked snapshot pass), 2-D tiles,
and device lists against the one-device container.  Not a test (pytest does not collect it; it uses the checker, so it lives under
tests/):  python tests/soak_r06.py [n]   (GPU box; prints one line per failure).  Round 6: 2000 rounds x 3, no failure."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
import orc as orc_mod
import llcomp_amd as mi
import test_gpu_stress as st

n = int(sys.argv[1]) if len(sys.argv) > 1 else 100
orc = orc_mod.Orc()
bad = 0
for i in range(n):
    for fn, base in ((st.run_chunked_case, 50000), (st.run_tiles_case, 60000)):
        try:
            fn(mi, orc, base + i)
        except AssertionError as e:
            bad += 1
            print("FAIL", fn.__name__, base + i, str(e)[:200], flush=True)
    rng = np.random.default_rng(70000 + i)
    w, h, c = int(rng.integers(8, 700)), int(rng.integers(2, 400)), int(rng.integers(1, 5))
    tw, th = int(rng.integers(1, w + 1)), int(rng.integers(1, min(h, 96) + 1))
    planar = bool(rng.integers(0, 2))
    img = st.make(rng, w, h, c, int(rng.integers(0, 5)))
    devs = [0] * int(rng.integers(2, 6))
    cpd = int(rng.integers(0, 6))
    try:
        one = mi.compress_image(img, w, h, c, format=mi.FORMAT_SLICED, tile_w=tw, tile_h=th, planar=planar, device=0)
        many = mi.compress_image(img, w, h, c, format=mi.FORMAT_SLICED, tile_w=tw, tile_h=th, planar=planar, devices=devs, chunks_per_device=cpd)
        assert many == one, "container"
        assert np.array_equal(mi.decompress_image(many, devices=devs, chunks_per_device=int(rng.integers(0, 6))).pixels, img), "pixels"
    except Exception as e:  # noqa: BLE001
        bad += 1
        print("FAIL devices", 70000 + i, (w, h, c, tw, th, planar, devs, cpd), repr(e)[:200], flush=True)
    if i % 20 == 19:
        print("...", i + 1, "rounds,", bad, "failures", flush=True)
print("soak done:", n, "rounds,", bad, "failures")
