#!/usr/bin/env python3
"""Markdown table of a bench matrix (profiles/rNNm_bench_matrix_10M.jsonl, one bench.py line per row).
usage: tools/matrix_table.py <matrix.jsonl> [<default bench line.json>]"""
import json
import re
import sys

rows = []
for f in sys.argv[1:]:
    for l in open(f):
        if l.strip():
            rows.append(json.loads(l))
print("| reference | reads | flags | form | reads/s (device path) | ms / step | kernels (ms): vote / seed / NW / prep | oracle sample | `abi_reads_per_s` | reference program, 16 thr | roofline frac (kernel) |")
print("|---|---|---|---|---|---|---|---|---|---|---|")
for j in rows:
    c = j["config"]; k = j["kernels"]
    flags = re.search(r"(-a 0\.9.*?), locate", c["workload"]).group(1).replace("NormalScoredSeq NW", "NW")
    g = lambda n: k.get(n, {}).get("ms_per_step", 0.0)
    cpu = j.get("cpu_baseline") or {}
    path = j.get("kernel_path", "")
    form = ("context records, " if "context records" in path else "") + (re.search(r"vote=(\S+)", path).group(1) if "vote=" in path else "-") + (" + k_seed" if "seeds=k_seed" in path else "")
    opts = " ".join(c.get("options") or [])
    ps = j.get("parity_sample") or {}
    print(f"| {c['genome_mbp']:g} Mbp{' repeat-rich' if c.get('repeat_rich') else ''} | {c['reads_per_gpu'] / 1e6:g} M x {c['read_len']} | `{flags}`{' ' + opts if opts else ''} | {form} | **{j['value'] / 1e6:.1f} M** | {j['ms_per_step']:.1f} | "
          f"{g('k_vote') + g('k_vote_retry'):.1f} / {g('k_seed'):.1f} / {g('k_nw'):.1f} / {g('k_prep'):.1f} | {ps.get('n', 0) - ps.get('mismatches', 0)} / {ps.get('n', 0)} | {(j.get('abi_reads_per_s') or 0) / 1e6:.1f} M | "
          f"{cpu.get('value', 0) / 1e3:.2f} k ({cpu.get('kind', '-')}) | {j['roofline']['frac']:.3f} ({j['roofline']['kernel']}) |")
