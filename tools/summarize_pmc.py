#!/usr/bin/env python3
"""Condense rocprofv3 CSVs (kernel_stats + counter_collection passes) into a per-kernel table."""
import csv
import glob
import os
import sys
from collections import defaultdict

out = sys.argv[1]
ks = os.path.join(out, "kernel_stats.csv")
if os.path.exists(ks):
    print("== kernel stats (rocprofv3 --kernel-trace --stats) ==")
    for row in csv.DictReader(open(ks)):
        name = row.get("Name", "")[:70]
        print("%-70s calls=%s avg_ns=%s min_ns=%s max_ns=%s pct=%s" % (
            name, row.get("Calls"), row.get("AverageNs"), row.get("MinNs"), row.get("MaxNs"), row.get("Percentage")))
print("== PMC (mean per dispatch) ==")
acc = defaultdict(lambda: defaultdict(list))
for f in sorted(glob.glob(os.path.join(out, "pmc*_counters.csv"))):
    for row in csv.DictReader(open(f)):
        k = row.get("Kernel_Name", "")[:60]
        acc[k][row.get("Counter_Name")].append(float(row.get("Counter_Value", 0)))
for k, ctrs in acc.items():
    if "molann" not in k and "frames" not in k and "mlp" not in k and "lane_jit" not in k and "chain" not in k:
        continue
    print(k)
    for c, v in sorted(ctrs.items()):
        print("   %-28s n=%-4d mean=%.6g" % (c, len(v), sum(v) / len(v)))
