#!/usr/bin/env python3
"""Condense a tools/prof.sh output directory: kernel stats + per-dispatch PMC means for the engine's kernels."""
import csv
import glob
import os
import sys
from collections import defaultdict

out = sys.argv[1]
for f in glob.glob(os.path.join(out, "trace", "**", "*kernel_stats.csv"), recursive=True):
    print("== kernel stats:", os.path.relpath(f, out))
    with open(f) as fh:
        for row in csv.DictReader(fh):
            print("  %-60s calls %6s  avg %12s ns  total %14s ns  %6s %%" % (
                row.get("Name", "")[:60], row.get("Calls"), row.get("AverageNs"), row.get("TotalDurationNs"), row.get("Percentage")))
for i in (1, 2, 3, 4):
    for f in glob.glob(os.path.join(out, f"pmc{i}", "**", "*counter_collection.csv"), recursive=True):
        acc = defaultdict(lambda: defaultdict(list))
        with open(f) as fh:
            for row in csv.DictReader(fh):
                acc[row["Kernel_Name"]][row["Counter_Name"]].append(float(row["Counter_Value"]))
        print("== PMC pass", i)
        for k, cs in acc.items():
            if "d2d_fir" not in k and "d2d_resample" not in k and "d2d_deinterleave" not in k:
                continue
            for c, v in sorted(cs.items()):
                print("  %-50s %-28s n=%4d mean=%.6g" % (k[:50], c, len(v), sum(v) / len(v)))
