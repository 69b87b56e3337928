"""Times the host-buffer calls (llcomp_mi_encode / llcomp_mi_decode) on one 4K frame: what a CLI user gets."""
import sys
import time

sys.path.insert(0, ".")
import numpy as np

import llcomp_amd as mi
from llcomp_amd.synth import gen_g3, gen_mid

W, H, C = 3840, 2160, 3
for name, gen in (("g3", gen_g3), ("mid", gen_mid)):
    img = gen(W, H, C)
    for tw, th, planar in ((480, 1, True), (64, 64, True)):
        s = mi.compress_image(img, W, H, C, format=mi.FORMAT_SLICED, tile_w=tw, tile_h=th, planar=planar)  # warm the codec cache
        assert np.array_equal(mi.decompress_image(s).pixels.reshape(img.shape), img)
        t0 = time.perf_counter()
        for _ in range(5):
            s = mi.compress_image(img, W, H, C, format=mi.FORMAT_SLICED, tile_w=tw, tile_h=th, planar=planar)
        t1 = time.perf_counter()
        for _ in range(5):
            r = mi.decompress_image(s)
        t2 = time.perf_counter()
        print(f"{name} {tw}x{th} planar={planar}: encode {(t1 - t0) / 5 * 1e3:.1f} ms, decode {(t2 - t1) / 5 * 1e3:.1f} ms per 4K frame "
              f"-> {W * H / 1e6 / ((t2 - t0) / 5):.0f} MPix/s enc+dec, ratio {img.size / len(s):.3f}", flush=True)
