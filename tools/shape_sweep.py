#!/usr/bin/env python3
"""Speed x compression-ratio table over slice shapes: 16 frames of 4K RGB8 per content, planar slices,
2 pipelines, a few steps each, through bench.measure (device-resident, bit-exact round trip checked inside).

    python tools/shape_sweep.py [out.txt]          # on a GPU box; appends one line per (content, shape)
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

SHAPES = [(480, 1), (3840, 1), (480, 2), (480, 4), (480, 8), (240, 16), (128, 32), (64, 64), (128, 128), (256, 256)]


def main():
    out = open(sys.argv[1], "a") if len(sys.argv) > 1 else sys.stdout
    import llcomp_amd as mi

    shifts = [None] if os.environ.get("SWEEP_SHIFTS") is None else [int(x) for x in os.environ["SWEEP_SHIFTS"].split(",")]
    for content in ("nat", "mid", "g3"):
        frames = bench.make_frames(content, 16, 0, distinct=4)
        for tw, th in SHAPES:
            for sh in shifts:
                if sh is not None:
                    os.environ["LLCOMP_MI_LANE_SHIFT"] = str(sh)
                    mi.reload_tuning()
                m = bench.measure(frames, tw, th, True, 2, 3, 1, 0)
                rec = {"content": content, "tile": f"{tw}x{th}", "lane_shift": sh, "mpix_s": round(m["mpix"], 1), "ratio": round(m["ratio"], 4),
                       "ms_per_step": round(m["dt"] / m["steps"] * 1e3, 2), "slices_per_frame": m["n_slices"] // 16,
                       "enc_ms": round(m["prof"]["k_encode_slices"] / m["steps"], 2), "dec_ms": round(m["prof"]["k_decode_slices"] / m["steps"], 2)}
                print(json.dumps(rec), file=out, flush=True)
                print(json.dumps(rec), flush=True)


if __name__ == "__main__":
    main()
