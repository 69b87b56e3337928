#!/usr/bin/env python3
"""BASELINE config 5 (128 noise frames of 4K host -> GPU -> host -> GPU -> host through llcomp_mi_stream_*): sweep of the
number of pipelines (stream objects, a driving thread each), slots, frames per job and encodes in flight.  bench.py's
c5 leg is one line of this table.     python tools/c5_sweep.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench

frames = bench.make_frames("g3", 32, 0, distinct=8)
for pipelines, depth, fpj, enc in ((1, 8, 4, 3), (1, 12, 4, 4), (2, 6, 4, 2), (2, 8, 4, 3), (2, 12, 4, 4), (2, 8, 2, 3), (3, 6, 4, 2), (4, 4, 4, 2), (4, 6, 4, 2)):
    try:
        r = bench.c5_stream(frames, 480, 1, True, depth=depth, frames_per_job=fpj, pipelines=pipelines, encodes_in_flight=enc)
        print(f"pipelines {pipelines} depth {depth} frames/job {fpj} encodes in flight {enc}: {r['value']:.0f} MPix/s steady, back-pressure {r['backpressure_hits']}", flush=True)
    except AssertionError as e:
        print(f"pipelines {pipelines} depth {depth} frames/job {fpj}: skipped ({e})", flush=True)
