#!/usr/bin/env python3
"""Summarise a rocprofv3 --kernel-trace --memory-copy-trace run of BASELINE config 5 (tools/c5_repeat.py under the profiler):
per copy direction the busy time (union of the copies' intervals), the summed time, bytes and the rate while busy, over the
middle 60 % of the run; the same for the library's kernels.   python tools/c5_trace_summary.py <rocprof output dir>"""
import csv
import glob
import sys

root = sys.argv[1]


def union(iv):
    iv = sorted(iv)
    busy, cur_s, cur_e = 0, None, None
    for s, e in iv:
        if cur_e is None or s > cur_e:
            if cur_e is not None:
                busy += cur_e - cur_s
            cur_s, cur_e = s, e
        else:
            cur_e = max(cur_e, e)
    if cur_e is not None:
        busy += cur_e - cur_s
    return busy


copies = {}
for f in glob.glob(root + "/**/*memory_copy_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        copies.setdefault(r["Direction"], []).append((int(r["Start_Timestamp"]), int(r["End_Timestamp"])))
kern = []
for f in glob.glob(root + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "llcomp_mi" in r["Kernel_Name"]:
            kern.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"])))
t0, t1 = min(s for s, e in kern), max(e for s, e in kern)   # the pipeline's run = first to last kernel of the library
lo, hi = t0 + 0.2 * (t1 - t0), t0 + 0.8 * (t1 - t0)
print(f"pipeline run {1e-6 * (t1 - t0):.1f} ms, window = its middle 60 % ({1e-6 * (hi - lo):.1f} ms)")
for d, v in sorted(copies.items()):
    w = [(max(s, lo), min(e, hi)) for s, e in v if e > lo and s < hi and e - s >= 50_000]   # the big copies (>= 50 us)
    busy = union(w)
    summed = sum(e - s for s, e in w)
    gaps = sorted((b[0] - a[1]) for a, b in zip(sorted(w), sorted(w)[1:]) if b[0] > a[1])
    print(f"{d:28s} copies >= 50 us: {len(w):4d}  busy {100 * busy / (hi - lo):5.1f} %  concurrency {summed / max(1, busy):4.2f}  "
          f"median copy {1e-6 * sorted(e - s for s, e in w)[len(w) // 2]:.2f} ms  idle gaps: {len(gaps)} totalling {1e-6 * sum(gaps):.1f} ms")
kw = [(max(s, lo), min(e, hi)) for s, e in kern if e > lo and s < hi]
print(f"library kernels: busy {100 * union(kw) / (hi - lo):.1f} %, concurrency {sum(e - s for s, e in kw) / max(1, union(kw)):.2f}")
