"""Times the legacy (single-stream, magic 0x79) path through the host-buffer calls on one GPU: DESIGN.md section 2."""
import sys, time
sys.path.insert(0, '.')
import numpy as np
import llcomp_amd as mi
from llcomp_amd.synth import gen_g2, gen_g3
for name, gen, w, h in (("g2", gen_g2, 1920, 1080), ("g3", gen_g3, 1920, 1080)):
    img = gen(w, h, 3)
    t = time.time(); s = mi.compress_image(img, w, h, 3); te = time.time() - t
    t = time.time(); r = mi.decompress_image(s); td = time.time() - t
    assert np.array_equal(r.pixels.reshape(img.shape), img)
    print(name, w, h, "legacy enc %.2fs dec %.2fs  -> %.3f MPix/s enc+dec" % (te, td, w*h/1e6/(te+td)), flush=True)
