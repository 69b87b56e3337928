#!/usr/bin/env python3
"""Not collected by pytest (a one-off for a GPU box, about a minute): full-size parity of the 2-D tile path -- 4K frames of five
contents x six tile shapes (planar and interleaved, the snapshot pass's capacity classes, ragged tiles) and a four-channel image,
every container equal to the oracle's byte for byte and decoded back to the input.

    python tests/big_parity.py
"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np, llcomp_amd as mi, orc as orc_mod
from llcomp_amd import synth
orc=orc_mod.Orc()
t0=time.time()
n=0
for gen in ("nat","mid","g3","g2","checker"):
    img=synth.GENERATORS[gen](3840,2160,3)
    for tw,th,planar in ((480,8,True),(128,32,True),(32,32,False),(240,16,True),(480,2,True),(61,67,True)):
        want=orc.compress_sliced(img,tw,th,planar)
        got=mi.compress_image(img,3840,2160,3,format=mi.FORMAT_SLICED,tile_w=tw,tile_h=th,planar=planar)
        assert got==want,(gen,tw,th,planar)
        assert np.array_equal(mi.decompress_image(got).pixels,img)
        n+=1
    print(gen,"ok",round(time.time()-t0,1),flush=True)
img4=np.concatenate([synth.gen_nat(2000,1200,3), synth.gen_g3(2000,1200,1)],axis=2)
for tw,th,planar in ((64,64,True),(32,32,False),(100,10,False)):
    want=orc.compress_sliced(img4,tw,th,planar); got=mi.compress_image(img4,2000,1200,4,format=mi.FORMAT_SLICED,tile_w=tw,tile_h=th,planar=planar)
    assert got==want; assert np.array_equal(mi.decompress_image(got).pixels,img4); n+=1
print("big parity:",n,"containers equal the oracle's, round trips lossless")
