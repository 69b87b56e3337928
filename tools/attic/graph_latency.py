#!/usr/bin/env python3
"""One 4K frame, encode + decode: the codec object's launch sequence issued call by call vs captured once into a HIP
graph (torch.cuda.CUDAGraph: the library launches on the capturing stream, nothing in it synchronises) and replayed.
    python tools/graph_latency.py [tile_w] [frames]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import llcomp_amd as mi
from llcomp_amd import synth

tw = int(sys.argv[1]) if len(sys.argv) > 1 else 480
F = int(sys.argv[2]) if len(sys.argv) > 2 else 1
W, H, C = 3840, 2160, 3
dev = torch.device("cuda", 0)
frames = np.stack([synth.gen_g3(W, H, C, seed=1234 + i) for i in range(F)])
px = torch.from_numpy(frames).to(dev)
out = torch.empty_like(px)
cd = mi.Codec(F, W, H, C, tw, 1, True, device=0)
cap = min(cd.max_payload_bytes, 2 * px.numel() + 64 * cd.n_slices + 4096)
payload = torch.empty(cap, dtype=torch.uint8, device=dev)
lens = torch.empty(cd.n_slices, dtype=torch.int32, device=dev)
total = torch.zeros(1, dtype=torch.int64, device=dev)
st = torch.zeros(2, dtype=torch.int32, device=dev)
s = torch.cuda.Stream(device=dev)


def both(stream_handle, nbytes):
    cd.encode(px.data_ptr(), payload.data_ptr(), cap, lens.data_ptr(), total.data_ptr(), st.data_ptr(), stream_handle)
    cd.decode(payload.data_ptr(), nbytes, lens.data_ptr(), out.data_ptr(), st[1:].data_ptr(), stream_handle)


with torch.cuda.stream(s):
    cd.encode(px.data_ptr(), payload.data_ptr(), cap, lens.data_ptr(), total.data_ptr(), st.data_ptr(), s.cuda_stream)
s.synchronize()
nbytes = int(total.item())
with torch.cuda.stream(s):
    both(s.cuda_stream, nbytes)
s.synchronize()
assert torch.equal(out, px) and int(st.sum().item()) == 0


def timed(fn, n=50):
    for _ in range(5):
        fn()
    s.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
        s.synchronize()  # latency: one frame at a time
    return (time.perf_counter() - t0) / n * 1e3


def direct():
    with torch.cuda.stream(s):
        both(s.cuda_stream, nbytes)


g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g, stream=s):
    both(torch.cuda.current_stream().cuda_stream, nbytes)


def replay():
    with torch.cuda.stream(s):
        g.replay()


a = timed(direct)
b = timed(replay)
out.zero_()
replay()
s.synchronize()
assert torch.equal(out, px)
print(f"{F} frame(s) 4K g3, {tw}x1 planar: launch by launch {a:.3f} ms, graph replay {b:.3f} ms per encode+decode")
