"""Times the host-buffer calls on one 4K frame: llcomp_mi_encode / llcomp_mi_decode with pageable buffers (what a CLI user
gets) and llcomp_mi_encode_into / llcomp_mi_decode_into with pinned buffers from llcomp_mi_host_alloc."""
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
        src, out, back = mi.PinnedBuffer(img.size), mi.PinnedBuffer(2 * img.size + (1 << 20)), mi.PinnedBuffer(img.size)
        src.array[:] = img.reshape(-1)
        kw = dict(format=mi.FORMAT_SLICED, tile_w=tw, tile_h=th, planar=planar)
        n = mi.compress_image_into(src.array, W, H, C, out.array, **kw)
        t0 = time.perf_counter()
        for _ in range(5):
            n = mi.compress_image_into(src.array, W, H, C, out.array, **kw)
        t1 = time.perf_counter()
        for _ in range(5):
            mi.decompress_image_into(out.array[:n], back.array)
        t2 = time.perf_counter()
        assert np.array_equal(back.array.reshape(img.shape), img)
        print(f"{name} {tw}x{th} planar={planar} PINNED: encode {(t1 - t0) / 5 * 1e3:.1f} ms, decode {(t2 - t1) / 5 * 1e3:.1f} ms per 4K frame "
              f"-> {W * H / 1e6 / ((t2 - t0) / 5):.0f} MPix/s enc+dec", flush=True)
        for b in (src, out, back):
            b.close()
