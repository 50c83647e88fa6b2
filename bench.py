#!/usr/bin/env python3
"""bench.py — reads/s of the MI355X seed-and-extend hot path through the C ABI of libgnumap_hip.so, on the workload
BASELINE.json quotes its metric on ("reads/sec (100 bp, -a 0.9) vs human ref"):

    default = human-scale synthetic reference (3.1 Gbp, 24 contigs, seed 42, 19 GB of index resident in HBM) + 10 M synthetic
    100-bp reads per GPU (1 % substitutions, 0.05 % insertions, 0.05 % deletions, Phred 20-40, 50 % reverse strand),
    -a 0.9 -m 14 -j 7 -k 2, no -h cap, NormalScoredSeq NW.
    Why -m 14 -j 7: the reference's default -m 10 means ~3000 SA hits per seed at 3.1 Gbp (10^5 per read; its own human script
    maps ONE read that way); 14 is the shortest seed whose expected chance hits per seed (11.5) stay below the true locus'
    votes, it needs no -h cap (so nothing is skipped) and keeps more sensitivity than the -m 20 -j 10 -h 150 SURVEY.md suggests.
    Other BASELINE configurations by flag:  configs[1]: --genome-mbp 100 --contigs 6 --mer 10 --jump 5
                                            configs[2] shape: --genome-mbp 156 --contigs 1 --mer 10 --jump 5 --max-kmer-hits 150
                                            configs[4] shape: --no-nw

`value`: a "step" is one pass of the device hot path (gm_map_batch_device: prep -> seed -> locate+vote -> NW -> hit compaction)
over the whole read set, resident in HBM before the timed region starts.  `abi_reads_per_s` (same JSON line) is the rate of the
path the ABI actually exports to a host driver: gm_map_batch + gm_output_batch on HOST buffers (upload, device path, unique-map
grouping, fp64 posterior pass on the host, traceback, CIGAR, SAM rows, coverage deposit, download), blocks of 1 M reads
driven by `--abi-threads` (2) host threads that keep `--abi-in-flight` (4) blocks queued each, one gm_batch + one HIP stream per
block in flight.  GPU_MAX_HW_QUEUES is raised to 16 for the process (unless the caller set it): the HIP runtime maps streams onto
4 hardware queues by default, where the kernels of blocks that share a queue run strictly one after another behind each other's
27-MB copies (measured on one box: 125 -> 140 M reads/s at 262 144-read blocks).

N > 1: one process per GPU (torch.distributed / RCCL), reads sharded with no data-path collective (weak scaling: per-GPU work
fixed); the per-position coverage track is the only thing that is all-reduced (once, in place, outside the per-step loop, like
the reference's MPI Allreduce at end of run); a known per-rank deposit pattern is checked after the reduction.

One JSON line on stdout (rank 0).  `roofline` is computed for the kernel with the largest device time, from algorithmic bytes
(kernel-side work counters x the per-unit byte costs of DESIGN.md) and HIP-event timings taken inside the library on the
launch stream.  `cpu_baseline` times the reference program itself (oracle/_ref/gnumap_ref, the unmodified src/Driver.cpp -c
pthread path; kind "reference") on a bounded sample of the same reads on this box's host cores, or the CPU restatement (kind
"port") when the reference binary did not travel.
"""
import argparse
import json
import os
import re
import subprocess
import sys
import threading
import time

os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")       # before anything initialises HIP (see the docstring)

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

HBM_PEAK_GBS = 8000.0       # MI355X_MICROARCH.md: HBM3E 8 TB/s spec


def log(*a):
    print(*a, file=sys.stderr, flush=True)


REPEAT_FAMILIES, REPEAT_LEN, REPEAT_FRACTION, REPEAT_DIVERGENCE = 1000, 300, 0.10, 0.05


def overlay_repeats(seq, rng, cons):
    """SURVEY.md 8(d) "repeat-rich" variant: 10 % of the bases come from 1 000 families of 300-bp elements, every copy with 5 %
    of its bases substituted.  `seq` holds codes 0..3 and is edited in place."""
    m = len(seq)
    k = int(REPEAT_FRACTION * m / REPEAT_LEN)
    if k == 0 or m <= REPEAT_LEN:
        return
    at = rng.integers(0, m - REPEAT_LEN, k)
    copies = cons[rng.integers(0, REPEAT_FAMILIES, k)]
    mut = rng.random(copies.shape) < REPEAT_DIVERGENCE
    copies = np.where(mut, (copies + rng.integers(1, 4, copies.shape, dtype=np.uint8)) & 3, copies).astype(np.uint8)
    seq[(at[:, None] + np.arange(REPEAT_LEN)[None, :]).reshape(-1)] = copies.reshape(-1)


def make_genome(path, mbp, seed, n_contigs, repeats=False):
    """i.i.d. uniform ACGT (optionally with the repeat overlay), split into contigs, FASTA with 100-column lines (deterministic),
    written in slabs"""
    rng = np.random.default_rng(seed)
    rrng = np.random.default_rng(seed + 1000)
    cons = rrng.integers(0, 4, (REPEAT_FAMILIES, REPEAT_LEN), dtype=np.uint8)
    G = int(mbp * 1_000_000)
    sizes = [G // n_contigs] * n_contigs
    sizes[-1] += G - sum(sizes)
    acgt = np.frombuffer(b"ACGT", np.uint8)
    with open(path, "wb") as f:
        for c, n in enumerate(sizes):
            f.write(b">chr%d synthetic seed %d\n" % (c + 1, seed))
            done = 0
            while done < n:
                m = min(n - done, 200_000_000)
                codes = rng.integers(0, 4, m, dtype=np.uint8)
                if repeats:
                    overlay_repeats(codes, rrng, cons)
                seq = acgt[codes]
                del codes
                rows = m // 100
                block = np.empty((rows, 101), np.uint8)
                block[:, :100] = seq[:rows * 100].reshape(rows, 100)
                block[:, 100] = 10
                f.write(block.tobytes())
                if m % 100:
                    f.write(seq[rows * 100:].tobytes() + b"\n")
                done += m


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


def algorithmic_bytes(c, L, n_reads, fused=False, bucket=False):
    """per-launch algorithmic bytes of each kernel from the kernel-side work counters (DESIGN.md 'Algorithmic bytes').
    fused: the seed lookup runs inside the vote kernel (no k_seed launch): k_prep also writes the reads' 2-bit forms (both strands +
    a header word), the vote kernel reads them and the table records instead of seed rows.
    bucket: the lookup is one k-mer -> positions record per k-mer: 4 bytes of header per probe + 4 bytes per located position (the
    positions ARE the suffix-array values; the record's unused words are not algorithmic bytes)"""
    win = L // 4 + 1
    forms = n_reads * (2 * (L // 4 + 1) + 4)
    vote = c["sa_hits"] * 4 + c["candidates"] * 16
    if bucket:
        vote += c.get("table_lookups", 0) * 4 + forms + 2 * n_reads * 6
    elif fused:
        vote += c["occ_blocks"] * 64 + c.get("table_lookups", 0) * 8 + forms + 2 * n_reads * 6
    else:
        vote += c["seeds_used"] * 12
    return {
        "k_prep": n_reads * (2 * L + 17) + (forms if fused else 0),
        "k_seed": c["occ_blocks"] * 64 + c.get("table_lookups", 0) * 8 + n_reads * 2 * L + c["seeds_used"] * 12 + 2 * n_reads * 6,
        "k_locate_sampled": c["lf_steps"] * 64 + c["sa_hits"] * 8,
        "k_vote": vote,
        "k_nw": c["candidates"] * (16 + 2 * L + win + 4),
        "k_compact(scan+scatter)": n_reads * 16 + c["candidates"] * 16 + c["accepted"] * 16,
        "k_vote_retry": 0,
    }


def write_fastq(path, B, Q, L, n):
    with open(path, "wb") as f:
        for s in range(0, n, 65536):
            e = min(n, s + 65536)
            f.write(b"".join(b"@r%d\n" % i + B[i, :L].tobytes() + b"\n+\n" + Q[i, :L].tobytes() + b"\n" for i in range(s, e)))


def cpu_baseline_reference(fa, B, Q, L, a, target_s, threads, wd):
    """the reference program itself (unmodified src/Driver.cpp etc., built by oracle/Makefile refbin) with -c <threads> on a bounded
    sample; mapping time = its own "Time since start" minus the same figure of a 1-read run (index load)"""
    exe = os.path.join(ROOT, "oracle", "_ref", "gnumap_ref")
    if not os.path.exists(exe):
        return None
    flags = ["-a", "0.9", "-m", str(a.mer), "-k", "2", "-c", str(threads), "-v", "1"]
    flags += ["-j", str(a.jump if a.jump else a.mer // 2)]
    if a.max_kmer_hits:
        flags += ["-h", str(a.max_kmer_hits)]
    if a.no_nw:
        flags += ["--no_nw"]
    fq = os.path.join(wd, "cpu_sample.fq")

    def run(n):
        write_fastq(fq, B, Q, L, n)
        r = subprocess.run([exe, "-g", fa, "-o", os.path.join(wd, "cpu_ref_out")] + flags + [fq], capture_output=True, text=True, timeout=1500)
        m = re.search(r"Time since start: ([0-9.eE+-]+)", r.stderr + r.stdout)
        if r.returncode != 0 or not m:
            raise RuntimeError(f"reference program failed (rc {r.returncode}): {(r.stderr or r.stdout)[-400:]}")
        return float(m.group(1))

    try:
        t_load = run(1)
        n = min(len(B), 256 * threads)
        t_map = 0.0
        for _ in range(4):
            t_map = max(run(n) - t_load, 1e-3)
            if t_map >= 0.5 * target_s or n >= len(B):
                break
            n = int(min(len(B), max(2 * n, n / t_map * target_s)))
    except Exception as e:                                    # the reference could not run here: say so, fall back to the port
        log(f"[bench] reference program not usable as CPU baseline: {e}")
        return None
    return dict(value=n / t_map, unit="reads/s", cores=threads, kind="reference", **host_cpu_info(),
                sample=f"first {n} of the benchmark reads through oracle/_ref/gnumap_ref (the unmodified reference program, -c {threads} pthreads, "
                       f"flags {' '.join(flags)}), mapping time {t_map:.1f} s = its 'Time since start' minus that of a 1-read run ({t_load:.1f} s of index load)",
                _n=n, _fq=fq, _sam=os.path.join(wd, "cpu_ref_out.sam"), _flags=[x for x in flags])


def host_cpu_info():
    """model string and core count of the box the CPU baseline ran on (BASELINE.md asks for both beside the rate)"""
    model = None
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                model = line.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    return dict(host_cpu=model, host_cores=os.cpu_count())


def sam_records(path):
    """the alignment lines of a SAM file, sorted (the reference with -c > 1 writes them in thread order)"""
    with open(path, "rb") as f:
        return sorted(l for l in f if not l.startswith(b"@"))


def parity_reference(cpu, fa, wd):
    """The reference-program run the CPU baseline just paid for IS a parity sample at the bench's own size: the same FASTQ (the first n
    benchmark reads) goes through the product's driver binary (gnumap_amd/bin/gnumap -> C ABI -> HIP kernels, default dispatch: the
    same kernel family the timed steps ran) with the same flags, and the two SAM record sets are compared BYTE FOR BYTE (sorted: the
    reference ran with -c > 1).  src/Driver.cpp:2146-2217 is what wrote the reference's side."""
    exe = os.path.join(ROOT, "gnumap_amd", "bin", "gnumap")
    if not cpu or cpu.get("kind") != "reference" or not os.path.exists(exe) or not os.path.exists(cpu["_sam"]):
        return None
    out = os.path.join(wd, "parity_mine")
    flags = [x for x in cpu["_flags"]]
    t0 = time.perf_counter()
    r = subprocess.run([exe, "-g", fa, "-o", out] + flags + [cpu["_fq"]], capture_output=True, text=True, timeout=900)
    if r.returncode != 0:
        raise SystemExit(f"[bench] parity_reference: the driver binary failed (rc {r.returncode}): {r.stderr[-600:]}")
    dt = time.perf_counter() - t0
    ref = sam_records(cpu["_sam"]); mine = sam_records(out + ".sam")
    bad = 0; first = None
    if ref != mine:
        sr, sm = set(ref), set(mine)
        diff = sorted(sr ^ sm)
        bad = len(diff)
        first = diff[0][:300].decode("latin1") if diff else "duplicate records differ"
        bad = bad or abs(len(ref) - len(mine)) or 1
    names = {l.split(b"\t", 1)[0] for l in ref}
    return dict(n=int(cpu["_n"]), records=len(ref), mapped_reads=len(names), mismatches=int(bad), first_difference=first, seconds=round(dt, 1),
                checked="every SAM record (QNAME FLAG RNAME POS MAPQ CIGAR SEQ QUAL XA XP X0) of the reference program's output on these reads, "
                        "byte for byte, against gnumap_amd/bin/gnumap with the same flags")


def cpu_baseline_port(fa, B, Q, L, kw, target_s, threads):
    """the oracle ("port": same algorithm and index layout as the reference, sampled SA + LF-walk locate) on a bounded sample"""
    from reflib import OracleLib
    orc = OracleLib()
    oix = orc.index_load(fa)
    op = orc.params(**kw)
    tmp = os.path.join(os.path.dirname(fa), "cpu_sample.fq")
    n = min(32 * threads, len(B))
    while True:                                  # grow the sample until the run is long enough to be a fair rate
        write_fastq(tmp, B, Q, L, n)
        st = orc.run(oix, op, tmp, "", threads=threads)
        if st.map_seconds >= 0.5 * target_s or n >= len(B):
            break
        rate = n / max(st.map_seconds, 1e-6)
        n = int(min(len(B), max(2 * n, rate * target_s)))
    return dict(value=n / st.map_seconds, unit="reads/s", cores=threads, kind="port", **host_cpu_info(),
                sample=f"first {n} of the benchmark reads, oracle/gm_oracle.c gmo_run with {threads} pthreads (mapping loop only, "
                       f"index preloaded, sampled-SA LF-walk locate as in the reference), {st.map_seconds:.1f} s")


def abi_rate(g, ix, p, B, Q, Ln, n_reads, block, n_threads, torch, in_flight=3, passes=3):
    """gm_map_batch + gm_output_batch on page-locked HOST buffers: `n_threads` caller threads, each keeping `in_flight` blocks queued
    through the enqueue / wait forms of the two calls (gm_map_batch_enqueue, gm_output_batch_enqueue, gm_batch_wait) on as many
    gm_batch + HIP stream pairs; in_flight = 1: the synchronous calls.  The blocks are gone through `passes` times (a 4 M-read leg is a
    40-ms measurement, +-40 % from run to run; three passes are steady state)"""
    n_reads = min(n_reads, len(B)) // block * block
    if n_reads == 0:
        return None
    Bp = g.api.pinned_empty((n_reads, B.shape[1]), np.uint8); Qp = g.api.pinned_empty((n_reads, B.shape[1]), np.uint8)
    Bp[:] = B[:n_reads]; Qp[:] = Q[:n_reads]
    Lp = g.api.pinned_empty(n_reads, np.uint16); Lp[:] = Ln[:n_reads]
    streams = [[torch.cuda.Stream() for _ in range(in_flight)] for _ in range(n_threads)]
    runners = [[g.api.BlockRunner(ix, p, block, B.shape[1], stream=s.cuda_stream) for s in ss] for ss in streams]
    blocks = list(range(0, n_reads, block))
    for rr in runners:                                          # warm-up: buffers sized, workspaces allocated
        for r in rr:
            r.run(Bp[:block], Qp[:block], Lp[:block])
    torch.cuda.synchronize()
    totals = [[0, 0] for _ in runners]
    err = []

    def work(k):
        try:
            mine = blocks[k::n_threads] * passes
            if in_flight == 1:
                for s in mine:
                    m, nr = runners[k][0].run(Bp[s:s + block], Qp[s:s + block], Lp[s:s + block])
                    totals[k][0] += m; totals[k][1] += nr
                return
            pending = [None] * in_flight
            for i, s in enumerate(mine):
                q = i % in_flight
                if pending[q] is not None:
                    m, nr = runners[k][q].wait(); totals[k][0] += m; totals[k][1] += nr
                runners[k][q].run_async(Bp[s:s + block], Qp[s:s + block], Lp[s:s + block]); pending[q] = s
            for q in range(in_flight):
                if pending[q] is not None:
                    m, nr = runners[k][q].wait(); totals[k][0] += m; totals[k][1] += nr
        except Exception as e:                                  # pragma: no cover
            err.append(e)

    th = [threading.Thread(target=work, args=(k,)) for k in range(n_threads)]
    t0 = time.perf_counter()
    for t in th:
        t.start()
    for t in th:
        t.join()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if err:
        raise err[0]
    n_reads *= passes
    return dict(reads_per_s=n_reads / dt, reads=n_reads, block=block, host_threads=n_threads, blocks_in_flight_per_thread=in_flight, seconds=dt,
                matches=sum(t[0] for t in totals), sam_records=sum(t[1] for t in totals))


def parity_sample(g, ix, p, fa, B, Q, Ln, L, kw, n_sample, block):
    """Outside the timed region: one block of the benchmark reads through gm_map_batch (the same kernels the timed steps ran - the
    kernel choice does not depend on the number of reads) and `n_sample` of its reads, spread over the block, compared with the
    oracle read by read: status, denominator, top score, every match's score bits and position set.  The oracle is the checker
    only (oracle/libgm_oracle.so); None when it did not travel."""
    try:
        from reflib import OracleLib
        orc = OracleLib()
    except Exception as e:                                    # pragma: no cover
        log(f"[bench] parity sample skipped: {e}")
        return None
    n = min(block, len(B))
    bt = g.Batch(ix, n, B.shape[1])
    res = bt.map(p, B[:n], Q[:n], Ln[:n])
    oix = orc.index_load(fa)
    op = orc.params(**kw)
    pick = np.unique(np.linspace(0, n - 1, n_sample).astype(np.int64))
    mb, M, P = res["match_begin"], res["matches"], res["positions"]
    bad = 0; n_matches = 0; unmapped = 0
    for i in pick:
        seq = B[i, :L].tobytes(); qual = Q[i, :L].tobytes()
        o = orc.map_read(oix, op, orc.pwm(seq, qual), seq)
        ms = M[int(mb[i]):int(mb[i + 1])]
        same = res["status"][i] == o["status"] and res["denominator"][i] == o["denominator"] and res["top_score"][i] == o["top_score"] and len(ms) == len(o["hits"])
        if same:
            for m, hh in zip(ms, o["hits"]):
                same &= bool(np.float32(m["score"]).view(np.uint32) == np.float32(hh["score"]).view(np.uint32))
                same &= [(int(q["pos"]), int(q["strand"])) for q in P[m["pos_begin"]:m["pos_end"]]] == [(int(x), int(y)) for x, y in hh["pos"]]
        n_matches += len(o["hits"]); unmapped += len(o["hits"]) == 0
        bad += not same
    bt.destroy()
    return dict(n=int(len(pick)), mismatches=int(bad), oracle_matches=int(n_matches), oracle_unmapped=int(unmapped), block=int(n),
                checked="status, denominator, top score, match score bits, position sets vs oracle/gm_oracle.c")


def reference_key(a):
    return f"g{a.genome_mbp:g}m_c{a.contigs}_s42" + ("r" if a.repeats else "")


def run_config(a, g, gd, torch, ix, fa, wd, B, Q, Ln, dev, rank, world, t_setup):
    """one flag set on the resident reference + reads: timed steps, ABI leg, parity sample, all-reduce, CPU baseline -> the JSON dict"""
    barrier = gd.barrier
    for kv in a.opt:
        g.set_option(*kv.split("=", 1))
    kw = dict(mer=a.mer, jump=a.jump, max_kmer_hits=a.max_kmer_hits, nw=0 if a.no_nw else 1, mode=a.mode)
    key = reference_key(a)
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
    kernel_path = batch.path()
    batch.set_profiling(False)

    # the path the ABI exports to a host driver (N = 1 only: it is a per-GPU figure and the host is shared)
    abi = None
    if a.abi_reads > 0 and world == 1:
        abi = abi_rate(g, ix, p, B, Q, Ln, a.abi_reads, a.abi_block, a.abi_threads, torch, a.abi_in_flight, a.abi_passes)
        log(f"[bench] ABI leg: {abi}")

    # self-check at the bench's own size: a sample of the benchmark reads against the oracle (rank 0, outside the timed region)
    parity = None
    if a.parity_sample > 0 and rank == 0:
        parity = parity_sample(g, ix, p, fa, B, Q, Ln, a.read_len, kw, a.parity_sample, min(a.abi_block, 262144))
        log(f"[bench] parity sample: {parity}")
        if parity and parity["mismatches"]:
            raise SystemExit(f"[bench] PARITY FAILURE: {parity}")

    # the one collective of the path: RCCL all-reduce of the device-resident coverage track, IN PLACE on the library's HBM buffer
    # (once per run, outside the per-step loop, like the reference's MPI Allreduce at end of run: src/Driver.cpp:1660-1672).  Every
    # rank first deposits the REAL coverage of one block of its own reads (gm_map_batch + gm_output_batch) plus a known pattern
    # (one common place, one own place); afterwards the pattern is checked bin by bin and the reduced track's total against the
    # sum of the per-rank totals.
    allreduce = None
    if world > 1 or os.environ.get("GM_FORCE_DIST") == "1":
        import torch.distributed as tdist
        ix.coverage_reset(p.bin_size)
        if p.mode:
            ix.coverage_enable_nuc()
        bins = ix.coverage_bins()
        nb = min(a.abi_block, 262144, len(B))
        bt = g.Batch(ix, nb, B.shape[1])
        res = bt.map(p, B[:nb], Q[:nb], Ln[:nb])
        recs, _ = bt.output(p, res)
        bt.destroy()
        bs = p.bin_size
        common, own = 8000, 800000 * (rank + 1)              # every rank deposits at `common`, and alone at `own`
        ix.coverage_add(np.array([common, own], np.uint64), np.array([64, 64], np.uint32), np.array([rank + 1.0, 1.0], np.float32))
        tracks = [("coverage", gd.DeviceTrack(ix.coverage_device_ptr(), bins).tensor(dev))]
        if p.mode:
            tracks.append(("per-nucleotide (5 x bins)", gd.DeviceTrack(ix.coverage_nuc_device_ptr(), 5 * bins).tensor(dev)))
        cov = tracks[0][1]
        torch.cuda.synchronize()
        # what the probed bins and the totals must become: the same sums taken over small copies / scalars
        probe = [common // bs + t for t in range(64 // bs)] + [800000 * (r + 1) // bs + t for r in range(world) for t in range(64 // bs)]
        probe_t = torch.tensor(probe, dtype=torch.int64, device=dev)
        want_probe = cov[probe_t].clone()
        if tdist.is_initialized():
            tdist.all_reduce(want_probe)
        own_totals = [float(t.sum(dtype=torch.float64).item()) for _, t in tracks]
        want_totals = [gd.sum_over_ranks(v, dev) for v in own_totals]
        torch.cuda.synchronize(); barrier()
        ta = time.perf_counter()
        for _, t in tracks:
            gd.allreduce_coverage(t)
        torch.cuda.synchronize()
        dt = gd.max_over_ranks(time.perf_counter() - ta, dev)
        got_totals = [float(t.sum(dtype=torch.float64).item()) for _, t in tracks]
        ok = all(abs(gt - wt) <= 1e-5 * max(1.0, abs(wt)) for gt, wt in zip(got_totals, want_totals))
        got_probe = cov[probe_t]
        ok = ok and bool(torch.allclose(got_probe, want_probe, rtol=1e-5, atol=1e-6))
        ok = ok and bool((got_probe[: 64 // bs] >= bs * world * (world + 1) / 2.0 - 1e-3).all()) and bool((got_probe[64 // bs:] >= float(bs) - 1e-3).all())
        if not ok:
            raise SystemExit(f"[bench] rank {rank}: coverage all-reduce gave a wrong track (totals {got_totals} vs {want_totals}; probe {got_probe[:4].tolist()} vs {want_probe[:4].tolist()})")
        nbytes = sum(t.numel() for _, t in tracks) * 4
        allreduce = dict(ms=dt * 1e3, bytes=nbytes, algbw_GBps=nbytes / dt / 1e9, busbw_GBps=nbytes / dt / 1e9 * 2 * (world - 1) / max(world, 1),
                         in_place=True, checked=True, tracks=[n_ for n_, _ in tracks],
                         deposit=f"gm_map_batch + gm_output_batch of {nb} reads per rank ({len(recs)} SAM records on rank 0) + a known pattern",
                         reduced_totals=got_totals, sum_of_rank_totals=want_totals)

    out = None
    if rank == 0:
        ms_per_step = elapsed / a.steps * 1e3
        value = world * a.reads * a.steps / elapsed
        fused = ktimes.get("k_seed", (0.0, 0))[1] == 0 and a.locate == "full"      # seed lookup inside the vote kernel
        alg = algorithmic_bytes(counters, a.read_len, a.reads, fused, "bucket" in kernel_path)
        per_kernel = {}
        for k, (ms, n) in ktimes.items():
            if n:
                # large batches run as sub-batches: launches per step = n / steps, each carrying 1/(n/steps) of the step's bytes
                lps = max(1, n // a.steps)
                per_kernel[k] = dict(ms=ms / n, ms_per_step=ms / a.steps, launches=n, alg_bytes=alg.get(k, 0) / lps,
                                     GBps=alg.get(k, 0) / (ms / a.steps * 1e-3) / 1e9)
        dom = max(per_kernel, key=lambda k: per_kernel[k]["ms_per_step"]) if per_kernel else None
        traffic = None; traffic_source = None
        pmc = os.path.join(ROOT, "profiles", "pmc_latest.json")
        wkey = f"{key}_L{a.read_len}_m{a.mer}_{a.locate}"
        if os.path.exists(pmc):
            try:
                j = json.load(open(pmc))
                # PMC passes are separate rocprofv3 runs of this command at 1 M reads per launch (tools/collect_profiles.sh); traffic scales
                # linearly with the reads of a launch
                if j.get("workload_key") == wkey and dom in j.get("kernels", {}):
                    lps = max(1, per_kernel[dom]["launches"] // a.steps)
                    traffic = int(j["kernels"][dom]["hbm_bytes_per_read"] * a.reads / lps)
                    traffic_source = (f"profiles/pmc_latest.json ({j.get('source', 'rocprofv3 --pmc passes')}: {j['kernels'][dom]['hbm_bytes_per_read']:.0f} B per read at "
                                      f"{j.get('reads_per_launch', '?')} reads per launch) x {a.reads // lps} reads per launch; NOT measured in this run")
            except Exception:
                traffic = None
        roof = None
        if dom:
            ach = per_kernel[dom]["GBps"]
            roof = dict(bound="hbm", kernel=dom, achieved=round(ach, 2), peak=HBM_PEAK_GBS, unit="GB/s", frac=round(ach / HBM_PEAK_GBS, 5),
                        traffic=traffic, traffic_source=traffic_source, avg_kernel_ms=round(per_kernel[dom]["ms"], 4),
                        alg_bytes_per_launch=int(per_kernel[dom]["alg_bytes"]))
        cpu = None
        if a.cpu_seconds > 0 and world == 1:         # rank 0 at N = 1 only
            threads = a.cpu_threads or min(16, os.cpu_count() or 1)
            if a.cpu_kind in ("auto", "reference"):
                cpu = cpu_baseline_reference(fa, B, Q, a.read_len, a, a.cpu_seconds, threads, wd)
            if cpu is None and a.cpu_kind != "reference":
                cpu = cpu_baseline_port(fa, B, Q, a.read_len, kw, a.cpu_seconds, threads)
        # the reference program's SAM of that same run, against the product's driver binary on the same FASTQ and flags
        parity_ref = None
        if cpu is not None and a.parity_reference:
            parity_ref = parity_reference(cpu, fa, wd)
            log(f"[bench] parity vs the reference program: {parity_ref}")
            if parity_ref and parity_ref["mismatches"]:
                raise SystemExit(f"[bench] PARITY FAILURE against the reference program: {parity_ref}")
        if cpu is not None:
            cpu = {k: v for k, v in cpu.items() if not k.startswith("_")}
        # which BASELINE.json configuration the run has the shape of
        if a.genome_mbp >= 3000 and not a.no_nw:
            shape = "the metric's configuration (human-scale reference, 100-bp reads, NormalScoredSeq NW)" if a.read_len == 100 else "human-scale reference"
        elif a.genome_mbp >= 3000:
            shape = "configs[4] shape (human-scale reference, --no_nw)"
        elif a.genome_mbp == 100.0 and a.mer == 10 and not a.no_nw:
            shape = "configs[1]"
        else:
            shape = "non-default workload"
        ref_kind = ("repeat-rich synthetic (10 % of the bases from 1000 families of 300-bp elements, 5 % divergence)" if a.repeats else "synthetic i.i.d.")
        out = {
            "metric": "reads/sec (100 bp, -a 0.9) vs human ref at 1/2/4/8 MI355X; HBM GB/s vs peak",
            "value": round(value, 1), "unit": "reads/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": round(ms_per_step, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u32 ranks + f32 scores", "data": "synthetic",
            "config": {"workload": f"{shape}: {ref_kind} {a.genome_mbp:g} Mbp reference ({a.contigs} contigs, seed 42) + {a.reads} x {a.read_len} bp reads per GPU, "
                                   f"-a 0.9 -m {p.mer} -j {p.jump} -k {p.min_seed_hits} -h {p.max_kmer_hits}, {'--no_nw' if a.no_nw else 'NormalScoredSeq NW'}, locate={a.locate}-SA",
                       "reads_per_gpu": a.reads, "read_len": a.read_len, "genome_mbp": a.genome_mbp, "repeat_rich": bool(a.repeats),
                       "sharding": f"reads x{world} (no data-path collective)"},
            "roofline": roof,
            "seed_lookup": "fused into the vote kernel" if fused else "k_seed",
            "kernel_path": kernel_path,
            "cpu_baseline": cpu,
            "parity_sample": parity,
            "parity_reference": parity_ref,
            "abi_reads_per_s": round(abi["reads_per_s"], 1) if abi else None,
            "abi": abi,
            "kernels": {k: {"ms_per_step": round(v["ms_per_step"], 4), "launches_per_step": v["launches"] // a.steps, "alg_GBps": round(v["GBps"], 2)} for k, v in per_kernel.items()},
            "counters_per_step": counters,
            "coverage_allreduce": allreduce,
            "pcie_inclusive_reads_per_s": round(pcie_inclusive, 1),
        }
    batch.destroy()
    for kv in a.opt:
        g.set_option(kv.split("=", 1)[0], None)
    if out is not None and a.opt:
        out["config"]["options"] = list(a.opt)
    return out


def build_parser():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--genome-mbp", type=float, default=3100.0)
    ap.add_argument("--contigs", type=int, default=24)
    ap.add_argument("--repeats", action="store_true", help="repeat-rich reference (SURVEY.md 8d): 10 %% of the bases from 1000 families of 300-bp elements, 5 %% divergence")
    ap.add_argument("--reads", type=int, default=10_000_000, help="reads per GPU")
    ap.add_argument("--read-len", type=int, default=100)
    ap.add_argument("--mer", type=int, default=14)
    ap.add_argument("--jump", type=int, default=0, help="0 = mer / 2 (the reference's default)")
    ap.add_argument("--max-kmer-hits", type=int, default=0)
    ap.add_argument("--no-nw", action="store_true")
    ap.add_argument("--mode", type=int, default=0, help="0 normal, 1 -b (bisulfite: bin size 1 and the 5 per-nucleotide tracks, all-reduced too), 2 --b2, 3 -d")
    ap.add_argument("--locate", choices=["full", "sampled"], default="full")
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="target wall time of the CPU baseline leg (0 = skip)")
    ap.add_argument("--cpu-threads", type=int, default=0)
    ap.add_argument("--cpu-kind", choices=["auto", "reference", "port"], default="auto")
    ap.add_argument("--abi-reads", type=int, default=8_388_608, help="reads of the gm_map_batch + gm_output_batch leg (0 = skip)")
    ap.add_argument("--abi-block", type=int, default=1048576)
    ap.add_argument("--abi-threads", type=int, default=2)
    ap.add_argument("--abi-passes", type=int, default=3, help="times the ABI leg goes through its blocks (more = a longer, steadier measurement)")
    ap.add_argument("--abi-in-flight", type=int, default=4, help="blocks each caller thread keeps queued (enqueue / wait forms); 1 = the synchronous calls")
    ap.add_argument("--parity-sample", type=int, default=64, help="reads of the benchmark compared with the oracle outside the timed region (0 = skip)")
    ap.add_argument("--parity-reference", type=int, default=1, help="compare the reference program's SAM of the CPU-baseline run with the driver binary's, byte for byte (0 = skip)")
    ap.add_argument("--workdir", default=os.environ.get("GM_BENCH_DIR", "/tmp/gnumap_bench"))
    ap.add_argument("--opt", action="append", default=[], metavar="GM_X=V", help="library run-time switch for this flag set (gm_set_option), e.g. --opt GM_SEED_FUSED=0")
    ap.add_argument("--also", action="append", default=[], metavar="FLAGS",
                    help="further flag sets measured in the same process on the same reference and reads (e.g. --also='--mer 20 --jump 10 "
                         "--max-kmer-hits 150'); one JSON line each, after the main one")
    return ap


def setup_reference(a, g, gd, torch, local):
    """reference + index, made once per box by local rank 0 BEFORE the process group exists (no rank waits inside a collective)"""
    key = reference_key(a)
    wd = os.path.join(a.workdir, key)
    fa = os.path.join(wd, "genome.fa")
    ready = fa + ".index_ready"
    if local == 0:
        os.makedirs(wd, exist_ok=True)
        if not os.path.exists(ready):
            log(f"[bench] generating {a.genome_mbp:g} Mbp {'repeat-rich ' if a.repeats else ''}reference and building its index (once; untimed)")
            make_genome(fa, a.genome_mbp, 42, a.contigs, a.repeats)
            g.index_build(fa)
            open(ready, "w").write("ok\n")
    else:
        while not os.path.exists(ready):
            time.sleep(1.0)
    return fa, wd


def main():
    ap = build_parser()
    a = ap.parse_args()

    import torch
    import gnumap_amd as g
    from gnumap_amd import dist as gd

    rank, world, local = gd.env_rank()
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: libgnumap_hip has no CPU fallback")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    t_setup = time.time()
    fa, wd = setup_reference(a, g, gd, torch, local)
    gd.init("nccl", dev)                                      # RCCL over xGMI when WORLD_SIZE > 1
    gd.barrier()
    flags = g.GM_INDEX_FULL_SA if a.locate == "full" else 0
    ix = g.Index(fa, device=local, flags=flags)
    # the packed reference as 2-bit codes for the read generator
    pac = np.fromfile(fa + ".gnumap.pac", np.uint8)[: ix.info.l_pac // 4 + 1]
    pac_t = torch.from_numpy(pac).to(dev)
    codes_t = torch.stack([(pac_t >> 6) & 3, (pac_t >> 4) & 3, (pac_t >> 2) & 3, pac_t & 3], 1).reshape(-1)[: ix.info.l_pac]
    del pac_t, pac
    B, Q, Ln = make_reads(codes_t, a.reads, a.read_len, gd.read_seed(1000, rank), dev)
    del codes_t
    torch.cuda.empty_cache()
    out = run_config(a, g, gd, torch, ix, fa, wd, B, Q, Ln, dev, rank, world, t_setup)
    if rank == 0:
        print(json.dumps(out), flush=True)
    for extra in a.also:                                     # same reference (and its flags), same reads, other mapping flags
        import shlex
        base, skip = [], False                               # the main row's flags without its own --opt / --also
        for x in sys.argv[1:]:
            if skip:
                skip = False
            elif x in ("--opt", "--also"):
                skip = True
            elif not (x.startswith("--opt=") or x.startswith("--also=")):
                base.append(x)
        a2 = ap.parse_args(base + shlex.split(extra))
        a2.also = []
        for fixed in ("genome_mbp", "contigs", "repeats", "reads", "read_len", "locate", "workdir"):
            setattr(a2, fixed, getattr(a, fixed))
        o2 = run_config(a2, g, gd, torch, ix, fa, wd, B, Q, Ln, dev, rank, world, time.time())
        if rank == 0:
            print(json.dumps(o2), flush=True)
    gd.shutdown()


if __name__ == "__main__":
    main()
