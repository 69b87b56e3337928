import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
import numpy as np
import orc as O
import llcomp_amd as mi
orc = O.Orc()
for (w, h, c) in [(1, 1, 1), (2, 1, 1), (3, 1, 1), (8, 1, 1), (1, 1, 3), (2, 1, 3), (40, 1, 1)]:
    img = O.gen_g1(w, h, c)
    s = orc.compress_image(img)
    try:
        out = mi.decompress_image(s).pixels
        exp = orc.forward_rct(img).reshape(-1)
        print((w, h, c), "ok" if np.array_equal(out, img) else "MISMATCH", "got", out.reshape(-1)[:12], "want", img.reshape(-1)[:12], "rct", exp[:12])
    except mi.LlcompError as e:
        print((w, h, c), "error", e)
