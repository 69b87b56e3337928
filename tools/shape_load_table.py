#!/usr/bin/env python3
"""One table out of tools/prof_shape_load.sh output directories: per slicing at load, the step's rate, every kernel's average launch
duration (kernel trace) and its HBM-side bytes per sample (2 x FETCH_SIZE + WRITE_SIZE, separate --pmc passes, KiB -> bytes).
    python tools/shape_load_table.py gpurun_out/r06/load_* > profiles/r06_shape_load.txt"""
import csv
import glob
import json
import os
import re
import sys
from collections import defaultdict


def per_kernel(root, sub, counter):
    acc, cnt = defaultdict(float), defaultdict(int)
    for f in glob.glob(os.path.join(root, sub, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            if row["Counter_Name"] == counter:
                m = re.search(r"(k_[a-z_0-9]+)(<[^>]*>)?", row["Kernel_Name"])
                k = (m.group(1) + (m.group(2) or "")) if m else row["Kernel_Name"][:40]
                acc[k] += float(row["Counter_Value"])
                cnt[k] += 1
    return {k: acc[k] / cnt[k] for k in acc}


for d in sys.argv[1:]:
    b = json.loads(open(os.path.join(d, "bench.json")).read().strip().splitlines()[-1])
    cfg = b["config"]
    samples = cfg["frames_per_step_per_gpu"] / cfg["streams"] * 3840 * 2160 * 3  # per launch
    print(f"== {os.path.basename(d)}: {cfg['workload']}")
    print(f"   {b['value']:.0f} MPix/s enc+dec, {b['ms_per_step']} ms per step, ratio {cfg['compression_ratio']}; library events per step (ms): "
          + ", ".join(f"{k} {v:.1f}" for k, v in b["kernel_ms_per_step"].items()))
    fetch, write = per_kernel(d, "pmc_fetch", "FETCH_SIZE"), per_kernel(d, "pmc_write", "WRITE_SIZE")
    dur = {}
    for f in glob.glob(os.path.join(d, "trace", "**", "*kernel_stats.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            m = re.search(r"(k_[a-z_0-9]+)(<[^>]*>)?", row["Name"])
            if m:
                dur[m.group(1) + (m.group(2) or "")] = (float(row["AverageNs"]) / 1e6, int(row["Calls"]))
    tot = 0.0
    # a kernel that is dispatched several times per encode / decode call (the chunked snapshot pass: once per chunk) counts that often
    base_calls = max([v[1] for k_, v in dur.items() if k_.startswith("k_model_fwd")] or [1])
    for k in sorted(fetch, key=lambda k_: -(2 * fetch[k_] + write.get(k_, 0))):
        ms = dur.get(k, (float("nan"), 0))
        times = max(1, round(ms[1] / base_calls)) if ms[1] and k.startswith("k_") and not k.startswith("k_scan_groups") else 1
        by = (2 * fetch[k] + write.get(k, 0.0)) * 1024 * times
        tot += by
        print(f"   {k:<62s} {ms[0]:8.3f} ms/dispatch under trace ({ms[1]:3d} dispatches, {times} per call)   {by / 1e9:7.2f} GB/call = {by / samples:6.1f} B/sample")
    print(f"   all kernels of one encode + one decode launch: {tot / 1e9:.1f} GB = {tot / samples:.1f} B/sample\n")
