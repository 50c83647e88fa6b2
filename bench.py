#!/usr/bin/env python3
"""bench.py — reads/s of the MI355X seed-and-extend hot path (gm_map_batch_device through the C ABI of
libgnumap_hip.so) on the workload BASELINE.json quotes its metric on:

    configs[1]: C. elegans-scale synthetic reference (100 Mbp, 6 contigs, seed 42) + 10 M synthetic 100-bp reads per GPU
    (1 % substitutions, 0.05 % insertions, 0.05 % deletions, Phred 20-40, 50 % reverse strand), gnumap defaults at -a 0.9
    (-m 10 -j 5 -k 2, no -h cap), NormalScoredSeq NW scoring.

A "step" is one pass of the hot path (prep -> seed -> locate+vote -> NW -> hit compaction) over the whole read set,
which is resident in HBM before the timed region starts.  N > 1: one process per GPU (torch.distributed / RCCL), reads
sharded with no data-path collective (weak scaling: per-GPU work fixed); the per-position coverage track is the only
thing that is all-reduced (once, outside the per-step loop, like the reference's MPI Allreduce at end of run).

One JSON line on stdout (rank 0).  `roofline` is computed for the kernel with the largest device time, from
algorithmic bytes (kernel-side work counters x the per-unit byte costs of DESIGN.md) and HIP-event timings taken
inside the library on the launch stream.  `cpu_baseline` times the CPU restatement (oracle/, "port") on a bounded
sample of the same reads with the reference's own data structures (sampled SA, LF-walk locate).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

HBM_PEAK_GBS = 8000.0       # MI355X_MICROARCH.md: HBM3E 8 TB/s spec


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def make_genome(path, mbp, seed, n_contigs):
    """i.i.d. uniform ACGT, split into contigs, FASTA with 100-column lines (deterministic)"""
    rng = np.random.default_rng(seed)
    G = int(mbp * 1_000_000)
    codes = rng.integers(0, 4, G, dtype=np.uint8)
    sizes = [G // n_contigs] * n_contigs
    sizes[-1] += G - sum(sizes)
    acgt = np.frombuffer(b"ACGT", np.uint8)
    with open(path, "wb") as f:
        off = 0
        for c, n in enumerate(sizes):
            f.write(b">chr%d synthetic seed %d\n" % (c + 1, seed))
            seq = acgt[codes[off:off + n]]
            rows = n // 100
            block = np.empty((rows, 101), np.uint8)
            block[:, :100] = seq[:rows * 100].reshape(rows, 100)
            block[:, 100] = 10
            f.write(block.tobytes())
            if n % 100:
                f.write(seq[rows * 100:].tobytes() + b"\n")
            off += n
    return codes


def make_reads(codes_t, n, L, seed, device):
    """synthetic reads on the GPU (torch is plumbing here), returned as host uint8 arrays [n, stride]"""
    import torch
    gen = torch.Generator(device=device)
    gen.manual_seed(seed)
    G = codes_t.numel()
    stride = (L + 7) // 8 * 8
    B = np.zeros((n, stride), np.uint8)
    Q = np.zeros((n, stride), np.uint8)
    acgt = torch.tensor(list(b"ACGT"), dtype=torch.uint8, device=device)
    ar = torch.arange(L, device=device)
    chunk = 1_000_000
    for s in range(0, n, chunk):
        m = min(chunk, n - s)
        pos = torch.randint(0, G - L - 2, (m,), generator=gen, device=device)
        # at most one indel per read: insertion / deletion with probability L * 0.0005 each
        u = torch.rand(m, generator=gen, device=device)
        p_ev = L * 0.0005
        kind = torch.where(u < p_ev, 1, torch.where(u < 2 * p_ev, -1, 0))          # +1 deletion (skip a base), -1 insertion
        where = torch.randint(5, L - 5, (m,), generator=gen, device=device)
        shift = (ar[None, :] >= where[:, None]).long() * kind[:, None]
        idx = pos[:, None] + ar[None, :] + shift
        c = codes_t[idx].long()
        ins = (kind == -1)[:, None] & (ar[None, :] == where[:, None])
        c = torch.where(ins, torch.randint(0, 4, (m, L), generator=gen, device=device), c)
        sub = torch.rand(m, L, generator=gen, device=device) < 0.01
        c = torch.where(sub, (c + torch.randint(1, 4, (m, L), generator=gen, device=device)) % 4, c)
        rev = torch.rand(m, generator=gen, device=device) < 0.5
        c = torch.where(rev[:, None], 3 - c.flip(1), c)
        q = torch.randint(20, 41, (m, L), generator=gen, device=device) + 33
        B[s:s + m, :L] = acgt[c].cpu().numpy()
        Q[s:s + m, :L] = q.to(torch.uint8).cpu().numpy()
    Ln = np.full(n, L, np.uint16)
    return B, Q, Ln


def algorithmic_bytes(c, L, n_reads):
    """per-launch algorithmic bytes of each kernel from the kernel-side work counters (DESIGN.md 'Algorithmic bytes')"""
    win = L // 4 + 1
    return {
        "k_prep": n_reads * (2 * L + 17),
        "k_seed": c["occ_blocks"] * 64 + c.get("table_lookups", 0) * 8 + n_reads * 2 * L + c["seeds_used"] * 12 + 2 * n_reads * 6,
        "k_locate_sampled": c["lf_steps"] * 64 + c["sa_hits"] * 8,
        "k_vote": c["sa_hits"] * 4 + c["seeds_used"] * 12 + c["candidates"] * 16,
        "k_nw": c["candidates"] * (16 + 2 * L + win + 4),
        "k_compact(scan+scatter)": n_reads * 16 + c["candidates"] * 16 + c["accepted"] * 16,
        "k_vote_retry": 0,
    }


def cpu_baseline(fa, B, Q, Ln, L, kw, target_s, threads):
    """the oracle ("port": same algorithm and index layout as the reference, sampled SA + LF-walk locate) on a bounded sample"""
    from reflib import OracleLib
    orc = OracleLib()
    oix = orc.index_load(fa)
    op = orc.params(**kw)

    def write_fq(path, n):
        with open(path, "wb") as f:
            for i in range(n):
                f.write(b"@r%d\n" % i + B[i, :L].tobytes() + b"\n+\n" + Q[i, :L].tobytes() + b"\n")

    tmp = os.path.join(os.path.dirname(fa), "cpu_sample.fq")
    n = min(32 * threads, len(B))
    while True:                                  # grow the sample until the run is long enough to be a fair rate
        write_fq(tmp, n)
        st = orc.run(oix, op, tmp, "", threads=threads)
        if st.map_seconds >= 0.5 * target_s or n >= len(B):
            break
        rate = n / max(st.map_seconds, 1e-6)
        n = int(min(len(B), max(2 * n, rate * target_s)))
    return dict(value=n / st.map_seconds, unit="reads/s", cores=threads, kind="port",
                sample=f"first {n} of the benchmark reads, oracle/gm_oracle.c gmo_run with {threads} pthreads (mapping loop only, "
                       f"index preloaded, sampled-SA LF-walk locate as in the reference), {st.map_seconds:.1f} s")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--genome-mbp", type=float, default=100.0)
    ap.add_argument("--contigs", type=int, default=6)
    ap.add_argument("--reads", type=int, default=10_000_000, help="reads per GPU")
    ap.add_argument("--read-len", type=int, default=100)
    ap.add_argument("--mer", type=int, default=10)
    ap.add_argument("--jump", type=int, default=0)
    ap.add_argument("--max-kmer-hits", type=int, default=0)
    ap.add_argument("--no-nw", action="store_true")
    ap.add_argument("--locate", choices=["full", "sampled"], default="full")
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="target wall time of the CPU baseline leg (0 = skip)")
    ap.add_argument("--cpu-threads", type=int, default=0)
    ap.add_argument("--workdir", default=os.environ.get("GM_BENCH_DIR", "/tmp/gnumap_bench"))
    a = ap.parse_args()

    import torch
    import torch.distributed as dist
    import gnumap_amd as g
    from gnumap_amd import dist as gd

    rank, world, local = gd.env_rank()
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: libgnumap_hip has no CPU fallback")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    gd.init("nccl", dev)                                      # RCCL over xGMI when WORLD_SIZE > 1
    barrier = gd.barrier

    kw = dict(mer=a.mer, jump=a.jump, max_kmer_hits=a.max_kmer_hits, nw=0 if a.no_nw else 1)
    key = f"g{a.genome_mbp:g}m_c{a.contigs}_s42"
    wd = os.path.join(a.workdir, key)
    fa = os.path.join(wd, "genome.fa")
    t_setup = time.time()
    if local == 0:
        os.makedirs(wd, exist_ok=True)
        if not os.path.exists(fa + ".gnumap.sa"):
            log(f"[bench] generating {a.genome_mbp:g} Mbp reference and building its index (once; untimed)")
            make_genome(fa, a.genome_mbp, 42, a.contigs)
            g.index_build(fa)
    barrier()
    flags = g.GM_INDEX_FULL_SA if a.locate == "full" else 0
    ix = g.Index(fa, device=local, flags=flags)
    # the packed reference as 2-bit codes for the read generator
    pac = np.fromfile(fa + ".gnumap.pac", np.uint8)[: ix.info.l_pac // 4 + 1]
    codes = np.stack([(pac >> 6) & 3, (pac >> 4) & 3, (pac >> 2) & 3, pac & 3], 1).reshape(-1)[: ix.info.l_pac]
    codes_t = torch.from_numpy(codes).to(dev)
    B, Q, Ln = make_reads(codes_t, a.reads, a.read_len, gd.read_seed(1000, rank), dev)
    del codes_t
    torch.cuda.empty_cache()
    p = g.Params(**kw)
    batch = g.Batch(ix, a.reads, B.shape[1])
    batch.upload(p, B, Q, Ln)                   # reads resident in HBM from here on
    ix.coverage_reset(p.bin_size)
    torch.cuda.synchronize()
    # PCIe-inclusive rate (never `value`): host -> HBM upload of the batch + one pass, measured once outside the timed region
    t_up = time.perf_counter()
    batch.upload(p, B, Q, Ln)
    batch.map_device(p)
    torch.cuda.synchronize()
    pcie_inclusive = a.reads / (time.perf_counter() - t_up)
    log(f"[bench] rank {rank}: setup {time.time() - t_setup:.1f} s, index {ix.info.hbm_bytes / 1e9:.2f} GB in HBM, {a.reads} reads resident")

    for _ in range(a.warmup):
        batch.map_device(p)
    torch.cuda.synchronize()
    batch.kernel_times()                         # drop warm-up timings
    batch.set_profiling(True)
    barrier(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        batch.map_device(p)
    torch.cuda.synchronize(); barrier()
    t1 = time.perf_counter()
    elapsed = gd.max_over_ranks(t1 - t0, dev)
    ktimes = batch.kernel_times()
    counters = batch.counters()

    # the one collective of the path: RCCL all-reduce of the device-resident coverage track (once per run, outside the
    # per-step loop, like the reference's MPI Allreduce at end of run: src/Driver.cpp:1660-1672)
    allreduce_ms = None
    if world > 1 or os.environ.get("GM_FORCE_DIST") == "1":
        cov = gd.DeviceTrack(ix.coverage_device_ptr(), ix.coverage_bins()).tensor(dev)
        torch.cuda.synchronize(); barrier()
        ta = time.perf_counter()
        gd.allreduce_coverage(cov)
        torch.cuda.synchronize()
        allreduce_ms = (time.perf_counter() - ta) * 1e3

    if rank == 0:
        ms_per_step = elapsed / a.steps * 1e3
        value = world * a.reads * a.steps / elapsed
        alg = algorithmic_bytes(counters, a.read_len, a.reads)
        per_kernel = {}
        for k, (ms, n) in ktimes.items():
            if n:
                # large batches run as sub-batches: launches per step = n / steps, each carrying 1/(n/steps) of the step's bytes
                lps = max(1, n // a.steps)
                per_kernel[k] = dict(ms=ms / n, ms_per_step=ms / a.steps, launches=n, alg_bytes=alg.get(k, 0) / lps,
                                     GBps=alg.get(k, 0) / (ms / a.steps * 1e-3) / 1e9)
        dom = max(per_kernel, key=lambda k: per_kernel[k]["ms_per_step"]) if per_kernel else None
        traffic = None
        pmc = os.path.join(ROOT, "profiles", "pmc_latest.json")
        if os.path.exists(pmc):
            try:
                j = json.load(open(pmc))
                # PMC passes are taken at 1 M reads per launch; traffic scales linearly with the reads of a launch
                if j.get("workload_key") == f"{key}_L{a.read_len}_m{a.mer}_{a.locate}" and j.get("kernel") == dom:
                    traffic = int(j["hbm_bytes_per_read"] * a.reads / max(1, per_kernel[dom]["launches"] // a.steps))
            except Exception:
                traffic = None
        roof = None
        if dom:
            ach = per_kernel[dom]["GBps"]
            roof = dict(bound="hbm", kernel=dom, achieved=round(ach, 2), peak=HBM_PEAK_GBS, unit="GB/s", frac=round(ach / HBM_PEAK_GBS, 5),
                        traffic=traffic, avg_kernel_ms=round(per_kernel[dom]["ms"], 4), alg_bytes_per_launch=int(per_kernel[dom]["alg_bytes"]))
        cpu = None
        if a.cpu_seconds > 0 and world == 1:         # rank 0 at N = 1 only
            threads = a.cpu_threads or min(16, os.cpu_count() or 1)
            cpu = cpu_baseline(fa, B, Q, Ln, a.read_len, kw, a.cpu_seconds, threads)
        # which BASELINE.json configuration the run has the shape of (default flags = configs[1], the one the metric is quoted on)
        if a.genome_mbp == 100.0 and a.mer == 10 and not a.no_nw:
            shape = "configs[1]"
        elif a.genome_mbp >= 3000:
            shape = "configs[4] shape (human-scale reference, --no_nw)" if a.no_nw else "human-scale reference (configs[3]/[4] size, NormalScoredSeq)"
        else:
            shape = "non-default workload"
        out = {
            "metric": "reads/sec (100 bp, -a 0.9) vs human ref at 1/2/4/8 MI355X; HBM GB/s vs peak",
            "value": round(value, 1), "unit": "reads/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": round(ms_per_step, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u32 ranks + f32 scores", "data": "synthetic",
            "config": {"workload": f"{shape}: synthetic {a.genome_mbp:g} Mbp reference ({a.contigs} contigs, seed 42) + {a.reads} x {a.read_len} bp reads per GPU, "
                                   f"-a 0.9 -m {p.mer} -j {p.jump} -k {p.min_seed_hits} -h {p.max_kmer_hits}, {'--no_nw' if a.no_nw else 'NormalScoredSeq NW'}, locate={a.locate}-SA",
                       "reads_per_gpu": a.reads, "read_len": a.read_len, "genome_mbp": a.genome_mbp, "sharding": f"reads x{world} (no data-path collective)"},
            "roofline": roof,
            "cpu_baseline": cpu,
            "kernels": {k: {"ms_per_step": round(v["ms_per_step"], 4), "launches_per_step": v["launches"] // a.steps, "alg_GBps": round(v["GBps"], 2)} for k, v in per_kernel.items()},
            "counters_per_step": counters,
            "coverage_allreduce_ms": allreduce_ms,
            "pcie_inclusive_reads_per_s": round(pcie_inclusive, 1),
        }
        print(json.dumps(out), flush=True)
    gd.shutdown()


if __name__ == "__main__":
    main()
