#!/usr/bin/env python3
"""Time the resident hot path (bench.py's step: Batch.map_device) under several settings of the library's run-time switches in ONE
process, so that the reference, its index and the reads are made once:

    python3 tools/env_sweep.py --sets "" "GM_PIPELINE=1000000" "GM_PIPELINE=2500000 GM_VOTE_FIXED=0"

Only switches the library reads per call take effect (GM_PIPELINE, GM_VOTE, GM_VOTE_SLOTS, GM_HEAVY_MIN ...); the ones it latches
in a static on first use (GM_NW_ROWS, GM_VOTE_FIXED, GM_TRACEBACK ...) need a process of their own.  Prints one line per setting."""
import argparse, os, sys, time
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--genome-mbp", type=float, default=3100.0)
    ap.add_argument("--contigs", type=int, default=24)
    ap.add_argument("--reads", type=int, default=10_000_000)
    ap.add_argument("--read-len", type=int, default=100)
    ap.add_argument("--mer", type=int, default=14)
    ap.add_argument("--jump", type=int, default=0)
    ap.add_argument("--max-kmer-hits", type=int, default=0)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--sets", nargs="+", default=[""])
    ap.add_argument("--workdir", default=os.environ.get("GM_BENCH_DIR", "/tmp/gnumap_bench"))
    a = ap.parse_args()
    import torch
    import gnumap_amd as g
    dev = torch.device("cuda", 0)
    key = f"g{a.genome_mbp:g}m_c{a.contigs}_s42"
    wd = os.path.join(a.workdir, key); fa = os.path.join(wd, "genome.fa"); ready = fa + ".index_ready"
    os.makedirs(wd, exist_ok=True)
    if not os.path.exists(ready):
        bench.make_genome(fa, a.genome_mbp, 42, a.contigs)
        g.index_build(fa)
        open(ready, "w").write("ok\n")
    ix = g.Index(fa, device=0, flags=g.GM_INDEX_FULL_SA)
    pac = np.fromfile(fa + ".gnumap.pac", np.uint8)[: ix.info.l_pac // 4 + 1]
    pac_t = torch.from_numpy(pac).to(dev)
    codes_t = torch.stack([(pac_t >> 6) & 3, (pac_t >> 4) & 3, (pac_t >> 2) & 3, pac_t & 3], 1).reshape(-1)[: ix.info.l_pac]
    del pac_t, pac
    B, Q, Ln = bench.make_reads(codes_t, a.reads, a.read_len, 1000, dev)
    del codes_t
    torch.cuda.empty_cache()
    p = g.Params(mer=a.mer, jump=a.jump, max_kmer_hits=a.max_kmer_hits)
    batch = g.Batch(ix, a.reads, B.shape[1])
    batch.upload(p, B, Q, Ln)
    ix.coverage_reset(p.bin_size)
    base_counters = None
    for s in a.sets:
        sets = dict(kv.split("=", 1) for kv in s.split()) if s.strip() else {}
        for k, v in sets.items():
            os.environ[k] = v
        batch.map_device(p)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(a.steps):
            batch.map_device(p)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / a.steps * 1e3
        c = batch.counters()
        batch.set_profiling(True)                      # one more pass with HIP events around every kernel
        batch.kernel_times()
        batch.map_device(p)
        torch.cuda.synchronize()
        kt = batch.kernel_times()
        batch.set_profiling(False)
        per = "  ".join(f"{k.split('(')[0][2:]} {v[0]:.2f}" for k, v in kt.items() if v[1])
        same = True
        if base_counters is None:
            base_counters = c
        else:
            same = all(c[k] == base_counters[k] for k in ("candidates", "accepted", "nw_cells", "sa_hits"))
        print(f"[sweep] {s or '(default)':48s} {ms:8.3f} ms / step  {a.reads / ms / 1e3:8.1f} M reads/s  counters {'same' if same else 'DIFFER'}  | {per}", flush=True)
        for k in sets:
            del os.environ[k]


if __name__ == "__main__":
    main()
