#!/usr/bin/env python3
"""Small fixed workload for rocprofv3 --pmc passes over the slice kernels: 32 frames of the headline content, one pipeline,
three encode + decode round trips (verified).  LLCOMP_MI_LIB selects the build."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

frames = bench.make_frames(os.environ.get("EXP_CONTENT", "g3"), 32, 0, distinct=8)
m = bench.measure(frames, 480, 1, True, 1, 2, 1, 0)
print("mpix", round(m["mpix"], 1), flush=True)
