#!/usr/bin/env python3
"""BASELINE config 5 leg of bench.py, repeated in one process: how reproducible is it, what do verification, thread
placement and the pipeline's shape cost?   python tools/c5_repeat.py [repeats=3] [variant ...]"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
only = set(sys.argv[2:])
frames = bench.make_frames("g3", 32, 0, distinct=8)
print("link:", json.dumps(bench.link_rate()), " env:", {k: v for k, v in os.environ.items() if k.startswith(("HSA_", "GPU_", "HIP_"))}, flush=True)
local = bench.gpu_local_cpus(0)
print("gpu-local cpus:", len(local or []), "of", len(os.sched_getaffinity(0)), "usable", flush=True)
VARIANTS = (("default", {}), ("long: 8 passes", {"passes": 8}), ("long3: 8 passes 3 pipelines", {"passes": 8, "pipelines": 3, "frames_per_job": 8, "depth": 4}), ("no pinning", {"pin": False}), ("no verification", {"verify": False}),
            ("4 pipelines depth 6", {"pipelines": 4, "depth": 6, "encodes_in_flight": 2}),
            ("3 pipelines depth 6", {"pipelines": 3, "depth": 6, "encodes_in_flight": 2, "frames_per_job": 2}),
            ("2 pipelines 8 frames/job", {"frames_per_job": 8, "depth": 6, "encodes_in_flight": 2}),
            ("1 pipeline depth 12", {"pipelines": 1, "depth": 12, "encodes_in_flight": 4}))
for label, kw in VARIANTS:
    if only and label.split()[0] not in only and label not in only:
        continue
    vals = []
    for _ in range(reps):
        r = bench.c5_stream(frames, 480, 1, True, **kw)
        vals.append(r["value"])
    print(f"{label:26s}", " ".join(f"{v:7.0f}" for v in vals), f"  median {sorted(vals)[len(vals) // 2]:.0f}  spread {(max(vals) - min(vals)) / max(vals):.1%}", flush=True)
