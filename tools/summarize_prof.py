#!/usr/bin/env python3
"""Trim a rocprofv3 --kernel-trace --stats output to the kernels of libgnumap_hip and copy it under profiles/.
usage: tools/summarize_prof.py <kernel_stats.csv> <out.csv> [bench.json]"""
import csv
import json
import sys

src, dst = sys.argv[1], sys.argv[2]
rows = list(csv.DictReader(open(src)))
ours = [r for r in rows if r["Name"].startswith(("k_", "void k_"))]
tot = sum(int(r["TotalDurationNs"]) for r in ours)
with open(dst, "w", newline="") as f:
    w = csv.writer(f)
    w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "PercentOfGnumapKernels", "MinNs", "MaxNs", "StdDev"])
    for r in sorted(ours, key=lambda r: -int(r["TotalDurationNs"])):
        w.writerow([r["Name"], r["Calls"], r["TotalDurationNs"], r["AverageNs"], f"{100.0 * int(r['TotalDurationNs']) / tot:.2f}", r["MinNs"], r["MaxNs"], r["StdDev"]])
if len(sys.argv) > 3:
    j = json.loads(open(sys.argv[3]).read().strip().splitlines()[-1])
    with open(dst.replace(".csv", ".bench.json"), "w") as f:
        json.dump(j, f, indent=1)
print(open(dst).read())
