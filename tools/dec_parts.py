#!/usr/bin/env python3
"""Where a sample of the 2-D decoder spends its time (diagnostic build, make -C llcomp_amd/csrc probe).

    python tools/dec_parts.py [--frames 16] [--streams 1] [--content g3] [--cache 0|1]

Every wavefront of k_decode_slices<NCH,false,false[,CACHE]> sums the shader cycles (s_memtime) of the four parts of a
sample -- context arithmetic, bank fetch (issue -> data in registers), decoding, write-back + neighbour rotation -- with the
queue drained at every stamp (the parts are serialised: a decomposition of the dependent chain, not of the undisturbed
kernel).  Prints the median over wavefronts of cycles per sample per part."""
import argparse
import ctypes as C
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
PROBE = os.path.join(ROOT, "llcomp_amd", "libllcomp_mi_probe.so")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=16)
    ap.add_argument("--streams", type=int, default=1)
    ap.add_argument("--content", default="g3")
    ap.add_argument("--tile", type=int, default=64)
    ap.add_argument("--cache", type=int, default=1, help="0: LLCOMP_MI_NOCACHE=1")
    args = ap.parse_args()
    os.environ["LLCOMP_MI_LIB"] = PROBE
    os.environ["LLCOMP_MI_NOCACHE"] = "0" if args.cache else "1"
    import numpy as np

    import bench
    from llcomp_amd import _lib

    L = _lib.load()
    L.llcomp_mi_probe_read.restype = C.c_uint32
    L.llcomp_mi_probe_read.argtypes = [C.c_void_p]
    L.llcomp_mi_probe_read_parts.argtypes = [C.c_void_p]
    slots = L.llcomp_mi_probe_read(None)
    frames = bench.make_frames(args.content, args.frames, 0, distinct=min(8, args.frames))
    m = bench.measure(frames, args.tile, args.tile, True, args.streams, 3, 1, 0)
    buf = np.zeros((2, slots, 2), dtype=np.uint64)
    parts = np.zeros((slots, 4), dtype=np.uint64)
    L.llcomp_mi_probe_read(buf.ctypes.data)
    L.llcomp_mi_probe_read_parts(parts.ctypes.data)
    samples = args.tile * args.tile
    p = parts.astype(np.float64)
    ok = p.sum(axis=1) > 0
    med = np.median(p[ok], axis=0) / samples
    t, r = buf[1, :, 0].astype(np.float64), buf[1, :, 1].astype(np.float64)
    okw = r > 1000
    res = {"config": vars(args), "probe_build_mpix": round(m["mpix"], 1), "wavefronts": int(ok.sum()),
           "cycles_per_sample": {"context": round(float(med[0]), 1), "bank_fetch": round(float(med[1]), 1), "decode": round(float(med[2]), 1),
                                 "writeback_rotate": round(float(med[3]), 1), "sum": round(float(med.sum()), 1)},
           "median_wave_lifetime_us": round(float(np.median(r[okw])) / 100.0, 1) if okw.any() else None,
           "clock_ghz": round(float(np.median(t[okw] / r[okw] * 0.1)), 3) if okw.any() else None}
    print(json.dumps(res))


if __name__ == "__main__":
    main()
