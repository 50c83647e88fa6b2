#!/usr/bin/env python3
"""Per-dispatch counter values of rocprofv3 --pmc (in dispatch order) for kernels whose name starts with a prefix.
usage: tools/pmc_dispatch.py <dir> <kernel-prefix>"""
import collections, csv, glob, os, sys
root, pre = sys.argv[1], sys.argv[2]
rows = collections.defaultdict(dict)
for f in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"].replace("void ", "").split("(")[0]
        if n.startswith(pre):
            rows[(int(r["Dispatch_Id"]), n)][r["Counter_Name"]] = float(r["Counter_Value"])
names = sorted({c for v in rows.values() for c in v})
print("dispatch kernel " + " ".join(names))
for (d, n), v in sorted(rows.items()):
    print(d, n, " ".join(f"{v.get(c, 0):.4g}" for c in names))
