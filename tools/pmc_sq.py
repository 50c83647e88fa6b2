#!/usr/bin/env python3
"""Mean per-launch value of every counter rocprofv3 --pmc collected, per kernel (all passes under <dir>).
usage: tools/pmc_sq.py <dir> [kernel-prefix]"""
import collections, csv, glob, os, sys
root = sys.argv[1]; pre = sys.argv[2] if len(sys.argv) > 2 else "k_vote"
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"].replace("void ", "").split("(")[0]
        if n.startswith(pre):
            agg[n][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in sorted(agg.items()):
    print(k)
    for c, x in sorted(v.items()):
        print(f"    {c:28s} {sum(x) / len(x):16.4e}   ({len(x)} launches)")
