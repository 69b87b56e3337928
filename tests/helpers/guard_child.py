#!/usr/bin/env python3
"""Child process of tests/test_gpu_asm_guard.py: runs a fixed set of cases through WHICHEVER build of the library
LLCOMP_MI_LIB names and prints one JSON line {case: [container FNV-1a-64, decoded-pixels FNV-1a-64]}.

    python tests/helpers/guard_child.py cases     the stress set (general + one-row path) and the 4K goldens' slicing
    python tests/helpers/guard_child.py refuse    one encode that the build must refuse (prints the status)
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, os.path.join(ROOT, "tests"))


def fnv(mi, b):
    return mi.fnv1a64(b)


def main():
    import llcomp_amd as mi
    from llcomp_amd import synth

    assert mi.device_count() >= 1
    what = sys.argv[1] if len(sys.argv) > 1 else "cases"
    if what == "refuse":
        img = synth.gen_g3(960, 8, 3)
        try:
            mi.compress_image(img, 960, 8, 3, format=mi.FORMAT_SLICED, tile_w=480, tile_h=1, planar=True)
            print(json.dumps({"status": 0}))
        except mi.LlcompError as e:
            print(json.dumps({"status": int(e.status), "message": str(e)}))
        return
    import test_gpu_stress as st

    out = {}
    rng_cases = [(1000 + i) for i in range(int(os.environ.get("GUARD_CASES", "40")))]
    for seed in rng_cases:
        # the stress generators, without the oracle: both builds must agree with EACH OTHER here (each is compared with the
        # oracle by the suite proper)
        rng = np.random.default_rng(seed)
        w, h, c = int(rng.integers(1, 700)), int(rng.integers(1, 300)), int(rng.integers(1, 5))
        img = st.make(rng, w, h, c, int(rng.integers(0, 5)))
        for tw, th, planar in ((int(rng.integers(1, w + 1)), 1, True), (int(rng.integers(1, w + 1)), 1, False), (int(rng.integers(8, 97)), int(rng.integers(2, 65)), bool(seed & 1))):
            s = mi.compress_image(img, w, h, c, format=mi.FORMAT_SLICED, tile_w=tw, tile_h=th, planar=planar)
            px = mi.decompress_image(s).pixels
            assert np.array_equal(px, img), f"round trip {seed} {tw}x{th}"
            out[f"s{seed}-{tw}x{th}{'p' if planar else 'i'}"] = [fnv(mi, s), fnv(mi, px.tobytes())]
    for gen in ("g3", "mid", "nat"):  # the benchmarked slicing at full size: 4K, planar 480x1
        img = synth.GENERATORS[gen](3840, 2160, 3)
        s = mi.compress_image(img, 3840, 2160, 3, format=mi.FORMAT_SLICED, tile_w=480, tile_h=1, planar=True)
        px = mi.decompress_image(s).pixels
        assert np.array_equal(px, img)
        out[f"4k-{gen}-480x1p"] = [fnv(mi, s), fnv(mi, px.tobytes()), len(s)]
    print(json.dumps(out))


if __name__ == "__main__":
    main()
