#!/usr/bin/env python3
"""BASELINE config 5 leg of bench.py, repeated in one process: how reproducible is it, what do verification and thread
placement cost?   python tools/c5_repeat.py [repeats=3]"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
frames = bench.make_frames("g3", 32, 0, distinct=8)
print("link:", json.dumps(bench.link_rate()), flush=True)
print("gpu-local cpus:", sorted(bench.gpu_local_cpus(0) or []), "usable:", sorted(os.sched_getaffinity(0)), flush=True)
for label, kw in (("default", {}), ("no pinning", {"pin": False}), ("no verification", {"verify": False}), ("4 pipelines depth 6", {"pipelines": 4, "depth": 6, "encodes_in_flight": 2}),
                  ("1 pipeline depth 12", {"pipelines": 1, "depth": 12, "encodes_in_flight": 4})):
    vals = []
    for _ in range(reps):
        r = bench.c5_stream(frames, 480, 1, True, **kw)
        vals.append(r["value"])
    print(f"{label:24s}", " ".join(f"{v:7.0f}" for v in vals), f"  spread {(max(vals) - min(vals)) / max(vals):.1%}", flush=True)
