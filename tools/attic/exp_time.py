#!/usr/bin/env python3
"""Times the slice kernels of a diagnostic build (make -C llcomp_amd/csrc exp EXP=n) on bench.py's headline workload, one
pipeline (kernels alone on the GPU) and three: what does the kernel time respond to?

    python tools/exp_time.py <library path or 'product'> [label]
Prints one JSON line: per-launch ms of k_encode_slices / k_decode_slices alone, and the 3-pipeline MPix/s."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
lib = sys.argv[1] if len(sys.argv) > 1 else "product"
if lib != "product":
    os.environ["LLCOMP_MI_LIB"] = os.path.abspath(lib)
    if not os.environ.get("EXP_CHECK"):  # (diagnostic builds with wrong bytes on purpose; EXP_CHECK=1: this build must round-trip)
        os.environ["LLCOMP_BENCH_NOCHECK"] = "1"
import bench  # noqa: E402

content = os.environ.get("EXP_CONTENT", "g3")
frames = bench.make_frames(content, 32, 0, distinct=8)
out = {"lib": os.path.basename(lib), "label": sys.argv[2] if len(sys.argv) > 2 else "", "content": content}
m1 = bench.measure(frames, 480, 1, True, 1, 6, 2, 0)
out["alone_ms"] = {k: round(m1["prof"][k] / max(1, m1["n_enc"]), 4) for k in ("k_encode_slices", "k_decode_slices")}
out["alone_mpix"] = round(m1["mpix"], 1)
m3 = bench.measure(frames, 480, 1, True, 3, 10, 2, 0)
out["three_pipelines_mpix"] = round(m3["mpix"], 1)
print(json.dumps(out), flush=True)
