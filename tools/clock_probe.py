#!/usr/bin/env python3
"""In-kernel shader clock of k_encode_slices / k_decode_slices under bench.py's default load.

    make -C llcomp_amd/csrc probe && python tools/clock_probe.py [--seconds 3] [--out profiles/r03_inkernel_clock.json]

Method (/opt/skills/guides/MI355X_MICROARCH.md, DVFS notes): a DIAGNOSTIC build of the library (libllcomp_mi_probe.so,
-DLLMI_CLOCK_PROBE) stamps s_memtime (shader cycles) and s_memrealtime (100 MHz) in every wavefront when it starts coding and
when it is done, into a buffer nothing else reads.  This script drives bench.py's headline workload (32 frames 4K noise,
480x1 planar slices, 3 pipelines) back to back for >= `seconds`, then reads the stamps of the last launches:
clock = delta(s_memtime) / delta(s_memrealtime) x 100 MHz per wavefront, median over wavefronts.  The product library contains
no stamp; bench.py reads the JSON this writes and prices the VALU-issue roofline at the measured clock."""
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
    ap.add_argument("--seconds", type=float, default=3.0)
    ap.add_argument("--frames", type=int, default=32)
    ap.add_argument("--streams", type=int, default=3)
    ap.add_argument("--content", default="g3")
    ap.add_argument("--tile-w", type=int, default=480)
    ap.add_argument("--tile-h", type=int, default=1)
    ap.add_argument("--out", default="")
    args = ap.parse_args()
    if not os.path.exists(PROBE):
        raise SystemExit(f"{PROBE} is missing: make -C llcomp_amd/csrc probe")
    os.environ["LLCOMP_MI_LIB"] = PROBE
    import numpy as np

    import bench
    from llcomp_amd import _lib

    L = _lib.load()
    L.llcomp_mi_probe_read.restype = C.c_uint32
    L.llcomp_mi_probe_read.argtypes = [C.c_void_p]
    slots = L.llcomp_mi_probe_read(None)
    frames = bench.make_frames(args.content, args.frames, 0, distinct=min(8, args.frames))
    m = bench.measure(frames, args.tile_w, args.tile_h, True, args.streams, 20, 2, 0)  # calibrate the step time
    steps = max(20, int(args.seconds / (m["dt"] / m["steps"])))
    buf = np.zeros((2, slots, 2), dtype=np.uint64)
    L.llcomp_mi_probe_read(buf.ctypes.data)  # clear
    m = bench.measure(frames, args.tile_w, args.tile_h, True, args.streams, steps, 1, 0)
    L.llcomp_mi_probe_read(buf.ctypes.data)
    res = {"method": "delta s_memtime / delta s_memrealtime x 100 MHz per wavefront around the coding loop, diagnostic build (-DLLMI_CLOCK_PROBE), "
                     f"after {m['dt']:.1f} s of back-to-back steps; median over wavefronts of the last launches",
           "config": {"frames_per_step_per_gpu": args.frames, "tile_w": args.tile_w, "tile_h": args.tile_h, "planar": True, "content": args.content,
                      "streams": args.streams}, "probe_build_mpix": round(m["mpix"], 1)}
    for k, name in ((0, "k_encode_slices"), (1, "k_decode_slices")):
        t, r = buf[k, :, 0].astype(np.float64), buf[k, :, 1].astype(np.float64)
        ok = r > 1000  # > 10 us
        ghz = t[ok] / r[ok] * 0.1
        res[name] = {"clock_ghz": round(float(np.median(ghz)), 4), "p05": round(float(np.percentile(ghz, 5)), 4), "p95": round(float(np.percentile(ghz, 95)), 4),
                     "wavefronts": int(ok.sum()), "median_wave_lifetime_us": round(float(np.median(r[ok])) / 100.0, 1),
                     "median_shader_cycles_per_wave": int(np.median(t[ok]))}
    print(json.dumps(res, indent=1))
    if args.out:
        with open(args.out, "w") as f:
            json.dump(res, f, indent=1)
            f.write("\n")


if __name__ == "__main__":
    main()
