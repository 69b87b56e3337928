#!/usr/bin/env python3
"""c4_inprocess_devices over the number of concurrent callers and the device list (one GPU: lists repeat ordinal 0)."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench
import llcomp_amd as mi
for devs in ([0], [0, 0], [0, 0, 0]):
    for callers in (1, 2, 3, 4):
        if callers * len(devs) > 6:
            continue
        r = bench.c4_inprocess(devs, images=max(4, callers * 2), callers=callers, compare_one_device=False, steps=2)
        print(json.dumps({"devices": devs, "callers": callers, "value": r["value"], "ms_per_image": r["ms_per_image_encode_plus_decode"]}), flush=True)
