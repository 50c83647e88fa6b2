#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc passes (separate runs for FETCH_SIZE, WRITE_SIZE, TCC_HIT/MISS, TCC_EA0_RDREQ) of bench.py.
usage: tools/pmc_summary.py <dir with one sub-directory per pass> <reads per launch> <workload_key> <out.txt> <out.json>
HBM bytes per launch follow MI355X_MICROARCH.md 'HBM': gfx950 tallies 128-byte read requests as 64 bytes, so
bytes = 2 x FETCH_SIZE x 1024 + WRITE_SIZE x 1024."""
import collections
import csv
import glob
import json
import os
import sys

root, reads, key, out_txt, out_json = sys.argv[1], int(sys.argv[2]), sys.argv[3], sys.argv[4], sys.argv[5]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(os.path.join(root, "*", "*", "*counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]
        if n.startswith(("k_", "void k_")):
            agg[n.split("(")[0].replace("void ", "")][r["Counter_Name"]].append(float(r["Counter_Value"]))
lines = []
best = None
for k, v in sorted(agg.items()):
    m = {c: sum(x) / len(x) for c, x in v.items()}
    fetch = m.get("FETCH_SIZE", 0.0) * 1024; write = m.get("WRITE_SIZE", 0.0) * 1024
    hbm = 2 * fetch + write
    hit = m.get("TCC_HIT_sum", 0.0); miss = m.get("TCC_MISS_sum", 0.0)
    lines.append(f"{k:32s} FETCH_SIZE={fetch / 1e9:8.3f} GB (x2 = {2 * fetch / 1e9:8.3f})  WRITE_SIZE={write / 1e9:7.3f} GB  HBM~{hbm / 1e9:8.3f} GB"
                 f"  L2 hit={hit / max(1.0, hit + miss):5.3f}  EA_RDREQ={m.get('TCC_EA0_RDREQ_sum', 0):.3e}")
    if best is None or hbm > best[1]:
        best = (k, hbm)
open(out_txt, "w").write(f"# rocprofv3 --pmc passes, {reads} reads per launch, workload {key}\n" + "\n".join(lines) + "\n")
def bench_name(k):                  # the names bench.py reports kernel times under (gm_kernel_name)
    for pre, nm in (("k_vote_retry", "k_vote_retry"), ("k_vote", "k_vote"), ("k_nw", "k_nw"), ("k_scan", "k_compact(scan+scatter)"),
                    ("k_scatter", "k_compact(scan+scatter)")):
        if k.startswith(pre):
            return nm
    return k.split("<")[0]


per = collections.defaultdict(float)
sym = {}
for k, v in agg.items():
    m = {c: sum(x) / len(x) for c, x in v.items()}
    hbm = 2 * m.get("FETCH_SIZE", 0.0) * 1024 + m.get("WRITE_SIZE", 0.0) * 1024
    per[bench_name(k)] += hbm
    sym.setdefault(bench_name(k), []).append(k)
json.dump({"workload_key": key, "reads_per_launch": reads,
           "note": "HBM bytes per launch = 2 x FETCH_SIZE x 1024 + WRITE_SIZE x 1024 (MI355X_MICROARCH.md: gfx950 tallies 128-byte reads as 64), separate --pmc passes",
           "kernels": {k: {"hbm_bytes_per_launch": v, "hbm_bytes_per_read": v / reads, "symbols": sym[k]} for k, v in sorted(per.items())}},
          open(out_json, "w"), indent=1)
print(open(out_txt).read())
