#!/usr/bin/env python3
"""Summarise rocprofv3 CSV output (kernel stats + counter_collection) per kernel name."""
import csv
import glob
import sys
from collections import defaultdict

root = sys.argv[1]
for f in glob.glob(root + "/**/*kernel_stats.csv", recursive=True):
    print("==", f)
    for row in csv.DictReader(open(f)):
        print("  %-60s calls=%s total_ns=%s avg_ns=%s pct=%s" % (row["Name"][:60], row["Calls"], row["TotalDurationNs"], row["AverageNs"], row["Percentage"]))
for f in glob.glob(root + "/**/*counter_collection.csv", recursive=True):
    print("==", f)
    acc = defaultdict(lambda: defaultdict(float))
    cnt = defaultdict(int)
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"][:50]
        acc[k][row["Counter_Name"]] += float(row["Counter_Value"])
        cnt[(k, row["Counter_Name"])] += 1
    for k, d in acc.items():
        print("  ", k)
        for c, v in sorted(d.items()):
            print("      %-24s total=%.4g  per_dispatch=%.4g (n=%d)" % (c, v, v / cnt[(k, c)], cnt[(k, c)]))
