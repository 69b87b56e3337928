#!/usr/bin/env python3
"""Turn a tools/prof_round.sh output directory into profiles/<name>: kernel stats + per-kernel HBM-side traffic
(FETCH_SIZE / WRITE_SIZE in KiB per dispatch, separate --pmc passes) and a small JSON bench.py reads for
roofline.traffic.  Correction per MI355X_MICROARCH.md (HBM): FETCH_SIZE reads exactly 1/2 of the bytes on gfx950 --
calibrated here on kernels with a known read volume (k_model_rows_inv / k_from_lane_order read the int16 lane-order
array once: measured FETCH_SIZE = 0.50 x bytes) -- so bytes = 2*FETCH_SIZE*1024 + WRITE_SIZE*1024."""
import csv
import glob
import json
import os
import shutil
import sys
from collections import defaultdict

src, dst, tag = sys.argv[1], sys.argv[2], sys.argv[3]
os.makedirs(dst, exist_ok=True)


def per_kernel(pattern):
    acc, cnt = defaultdict(float), defaultdict(int)
    for f in glob.glob(os.path.join(src, pattern, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            k = row["Kernel_Name"]
            acc[(k, row["Counter_Name"])] += float(row["Counter_Value"])
            cnt[(k, row["Counter_Name"])] += 1
    return {k: acc[k] / cnt[k] for k in acc}


fetch, write, sq1 = per_kernel("pmc_fetch"), per_kernel("pmc_write"), per_kernel("pmc_sq1")
out = {}
for (k, c), v in fetch.items():
    if "llcomp_mi" not in k:
        continue
    import re
    m = re.search(r"(k_[a-z_0-9]+)", k)
    short = m.group(1) if m else k
    w = write.get((k, "WRITE_SIZE"), 0.0)
    out[short] = {"fetch_size_kib": v, "write_size_kib": w, "hbm_bytes_corrected": int(2 * v * 1024 + w * 1024)}
    if (k, "SQ_INSTS_VALU") in sq1:  # VALU wave-instructions per launch (bench.py: roofline.valu_issue)
        out[short]["valu_insts"] = int(sq1[(k, "SQ_INSTS_VALU")])
bench = json.loads(open(os.path.join(src, "bench.json")).read().strip().splitlines()[-1])
try:  # the tree the passes ran on (run this script right behind the GPU call, before anything else is committed)
    import subprocess
    here = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    commit = subprocess.check_output(["git", "-C", here, "rev-parse", "--short=12", "HEAD"], text=True).strip()
    dirty = bool(subprocess.check_output(["git", "-C", here, "status", "--porcelain", "--", "llcomp_amd/csrc"], text=True).strip())
    commit += "+uncommitted kernel sources" if dirty else ""
except Exception:  # noqa: BLE001
    commit = None
doc = {"profiled_at_commit": commit, "config": bench["config"], "bench_value": bench["value"], "kernel_ms_per_step": bench["kernel_ms_per_step"], "per_launch": out,
       "correction": "bytes = 2*FETCH_SIZE + WRITE_SIZE (KiB -> bytes); FETCH_SIZE = 1/2 of known read volume on this access pattern"}
json.dump(doc, open(os.path.join(dst, f"{tag}_traffic.json"), "w"), indent=1)
for f in glob.glob(os.path.join(src, "trace", "**", "*kernel_stats.csv"), recursive=True):
    shutil.copy(f, os.path.join(dst, f"{tag}_kernel_stats.csv"))
shutil.copy(os.path.join(src, "summary.txt"), os.path.join(dst, f"{tag}_pmc_summary.txt"))
shutil.copy(os.path.join(src, "bench.json"), os.path.join(dst, f"{tag}_bench.json"))
print(json.dumps(out, indent=1))
