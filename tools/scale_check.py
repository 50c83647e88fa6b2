#!/usr/bin/env python3
"""Scale check of the hot path on a large synthetic reference (up to human scale, > 2^31 positions): build the index with
gm_index_build, load it into HBM, map reads drawn from the whole coordinate range (half of them from the last 5 % of the
reference, where 32-bit signed arithmetic would wrap), compare a random sample read by read with the oracle (test
infrastructure, like the tests), check that exact reads recover their origin, and report the rate.

    python3 tools/scale_check.py --mbp 2200 --reads 1000000 --mer 10 --max-kmer-hits 150

Prints one JSON line.  --build-only stops after the index build (no GPU needed)."""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
ACGT = np.frombuffer(b"ACGT", np.uint8)
COMP = np.zeros(256, np.uint8)
for a_, c_ in zip(b"ACGT", b"TGCA"):
    COMP[a_] = c_


_T0 = time.time()


def log(*a):
    print(f"[{time.time() - _T0:7.1f} s]", *a, file=sys.stderr, flush=True)


# repeat families of --repeats: (name, unit length, fraction of the genome, per-copy divergence, tandem copies per insertion)
FAMILIES = (("SINE-like", 300, 0.10, 0.12, 1), ("LINE-like", 6000, 0.15, 0.05, 1), ("satellite", 171, 0.03, 0.02, 60))


def overlay_repeats(seq, rng, cons):
    """overwrite random places of `seq` (ASCII) with diverged copies of the family consensus sequences (in place)"""
    m = len(seq)
    for (name, ln, frac, div, tandem), c in zip(FAMILIES, cons):
        span = ln * tandem
        k = int(frac * m / span)
        if k == 0 or m <= span:
            continue
        at = rng.integers(0, m - span, k)
        copies = np.tile(np.tile(c, tandem), (k, 1))
        mut = rng.random(copies.shape) < div
        copies[mut] = ACGT[rng.integers(0, 4, int(mut.sum()))]
        seq[(at[:, None] + np.arange(span)[None, :]).reshape(-1)] = copies.reshape(-1)


def write_genome(fa, G, n_contigs, seed, repeats=False):
    """i.i.d. ACGT (optionally overlaid with repeat families) in 100-column FASTA, generated contig by contig in 64 Mbp
    pieces (bounded memory); returns contig offsets"""
    rng = np.random.default_rng(seed)
    cons = [ACGT[rng.integers(0, 4, ln)] for _, ln, _, _, _ in FAMILIES]
    sizes = [G // n_contigs // 100 * 100] * n_contigs
    sizes[-1] += G - sum(sizes)
    offs = [0]
    with open(fa, "wb") as f:
        for c, n in enumerate(sizes):
            f.write(b">chr%d\n" % (c + 1))
            done = 0
            while done < n:
                m = min(64_000_000, n - done)
                seq = ACGT[rng.integers(0, 4, m, dtype=np.uint8)]
                if repeats:
                    overlay_repeats(seq, rng, cons)
                rows = m // 100
                blk = np.empty((rows, 101), np.uint8)
                blk[:, :100] = seq[:rows * 100].reshape(rows, 100)
                blk[:, 100] = 10
                f.write(blk.tobytes())
                if m % 100:
                    f.write(seq[rows * 100:].tobytes() + b"\n")
                done += m
            offs.append(offs[-1] + n)
    return offs


def check_output(g, ix, p, orc, oix, op, B, Q, Ln, L, pick):
    """gm_map_batch + gm_output_batch on the picked reads: SAM records, coverage bins and (-b / -d) per-nucleotide bins against the
    oracle's gmo_read_output; the device tracks are read through zero-copy views at the touched bins only"""
    import torch
    from gnumap_amd import dist as gd
    pick = np.asarray(sorted(set(int(x) for x in pick)))
    Bs, Qs, Ls = np.ascontiguousarray(B[pick]), np.ascontiguousarray(Q[pick]), np.ascontiguousarray(Ln[pick])
    bs = 1 if p.mode else p.bin_size
    ix.coverage_reset(bs)
    if p.mode:
        ix.coverage_enable_nuc()
    bt = g.Batch(ix, len(pick), Bs.shape[1])
    res = bt.map(p, Bs, Qs, Ls)
    recs, cigars = bt.output(p, res)
    want_recs = []; cov = {}; nuc = {}
    for k, i in enumerate(pick):
        seq = B[i, :L].tobytes(); qual = Q[i, :L].tobytes()
        st, orecs, deps = orc.read_output(oix, op, orc.pwm(seq, qual), seq)
        for r in orecs:
            want_recs.append((k, int(r["pos"]), int(r["strand"]), int(r["chr_pos"]), int(r["mapq"]), r["cigar"], np.float32(r["a_score"]).view(np.uint32),
                              np.float32(r["post_prob"]).view(np.uint32), int(r["sim_matches"])))
        for pos, span, w, codes in deps:
            for t in range(span):
                b_ = (pos + t) // bs
                cov[b_] = np.float32(cov.get(b_, np.float32(0)) + np.float32(w))
                if codes is not None and codes[t] < 5:
                    nuc[(codes[t], b_)] = np.float32(nuc.get((codes[t], b_), np.float32(0)) + np.float32(w))
    got_recs = [(int(r["read"]), int(r["pos"]), int(r["strand"]), int(r["chr_pos"]), int(r["mapq"]), c, np.float32(r["a_score"]).view(np.uint32),
                 np.float32(r["post_prob"]).view(np.uint32), int(r["sim_matches"])) for r, c in zip(recs, cigars)]
    bad_recs = int(got_recs != want_recs)
    dev = torch.device("cuda", 0)
    bins = ix.coverage_bins()
    track = gd.DeviceTrack(ix.coverage_device_ptr(), bins).tensor(dev)
    keys = np.fromiter(cov.keys(), np.int64, len(cov))
    got = track[torch.from_numpy(keys).to(dev)].cpu().numpy()
    want = np.array([cov[k] for k in keys], np.float32)
    bad_cov = int((np.abs(got - want) > 1e-5 * np.maximum(1.0, np.abs(want))).sum())
    total_ok = abs(float(track.sum().item()) - float(want.astype(np.float64).sum())) <= 1e-3 * max(1.0, float(want.sum()))
    bad_nuc = 0
    if p.mode:
        nt = gd.DeviceTrack(ix.coverage_nuc_device_ptr(), 5 * bins).tensor(dev)
        nk = list(nuc.keys())
        idx = torch.tensor([c * bins + b_ for c, b_ in nk], dtype=torch.int64, device=dev)
        gotn = nt[idx].cpu().numpy()
        wantn = np.array([nuc[k] for k in nk], np.float32)
        bad_nuc = int((np.abs(gotn - wantn) > 1e-5 * np.maximum(1.0, np.abs(wantn))).sum())
        total_ok = total_ok and abs(float(nt.sum().item()) - float(wantn.astype(np.float64).sum())) <= 1e-3 * max(1.0, float(wantn.sum()))
    bt.destroy()
    return dict(output_reads=len(pick), output_records=len(want_recs), output_record_mismatch=bad_recs, coverage_bins_checked=len(cov),
                coverage_bin_mismatches=bad_cov, nuc_bins_checked=len(nuc), nuc_bin_mismatches=bad_nuc, track_totals_ok=bool(total_ok))


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--mbp", type=float, default=2200.0)
    ap.add_argument("--contigs", type=int, default=24)
    ap.add_argument("--reads", type=int, default=1_000_000)
    ap.add_argument("--read-len", type=int, default=100)
    ap.add_argument("--mer", type=int, default=10)
    ap.add_argument("--max-kmer-hits", type=int, default=0)
    ap.add_argument("--sample", type=int, default=64, help="reads compared with the oracle")
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--workdir", default="/tmp/gnumap_scale")
    ap.add_argument("--build-only", action="store_true")
    ap.add_argument("--repeats", action="store_true", help="overlay SINE-/LINE-/satellite-like repeat families (28 %% of the genome)")
    ap.add_argument("--no-nw", action="store_true")
    ap.add_argument("--keep", action="store_true", help="keep the FASTA and the index files in --workdir")
    ap.add_argument("--batch", type=int, default=0, help="map the reads in blocks of this many (0 = one block); repeat-rich references without -h need it")
    ap.add_argument("--jump", type=int, default=0)
    ap.add_argument("--mode", type=int, default=0, help="0 normal, 1 -b, 2 --b2, 3 -d")
    ap.add_argument("--check-output", type=int, default=0, help="reads whose gm_output_batch records and coverage / per-nucleotide deposits are compared with the oracle")
    a = ap.parse_args(argv)

    import gnumap_amd as g
    G = int(a.mbp * 1e6)
    L = a.read_len
    os.makedirs(a.workdir, exist_ok=True)
    fa = os.path.join(a.workdir, f"g{a.mbp:g}{'r' if a.repeats else ''}_c{a.contigs}.fa")
    t_gen = t_build = 0.0
    if a.keep and os.path.exists(fa + ".offs.npy") and os.path.exists(fa + ".gnumap.sa"):
        offs = [int(x) for x in np.load(fa + ".offs.npy")]               # kept by an earlier call in the same session
    else:
        t = time.time()
        offs = write_genome(fa, G, a.contigs, 11, a.repeats)
        t_gen = time.time() - t
        log(f"[scale] {G} bp FASTA written in {t_gen:.0f} s")
        t = time.time()
        g.index_build(fa)
        t_build = time.time() - t
        log(f"[scale] index built in {t_build:.0f} s")
        if a.keep:
            np.save(fa + ".offs.npy", np.asarray(offs, np.int64))
    out = dict(genome_bp=G, contigs=a.contigs, repeats=bool(a.repeats), fasta_write_s=round(t_gen, 1), index_build_s=round(t_build, 1))
    if a.build_only:
        print(json.dumps(out)); return out

    t = time.time()
    ix = g.Index(fa, flags=g.GM_INDEX_FULL_SA)            # returns with the index resident (gm_index_open synchronises)
    out.update(index_load_s=round(time.time() - t, 1), index_hbm_gb=round(ix.info.hbm_bytes / 1e9, 2), l_pac=int(ix.info.l_pac))
    log(f"[scale] index in HBM: {out['index_hbm_gb']} GB, loaded in {out['index_load_s']} s")

    # reads: exact substrings with 1 % substitutions; half of them from the last 5 % of the coordinate range
    pac = np.memmap(fa + ".gnumap.pac", np.uint8, "r")
    rng = np.random.default_rng(5)
    n = a.reads
    lo_tail = int(G * 0.95)
    pos = np.where(rng.random(n) < 0.5, rng.integers(lo_tail, G - L, n), rng.integers(0, G - L, n)).astype(np.int64)
    # keep every read inside one contig
    oa = np.asarray(offs)
    ci = np.searchsorted(oa, pos, side="right") - 1
    pos = np.minimum(pos, oa[ci + 1] - L)
    strand = rng.integers(0, 2, n).astype(np.uint8)
    stride = (L + 7) // 8 * 8
    B = np.zeros((n, stride), np.uint8); Q = np.zeros((n, stride), np.uint8)
    is_exact = np.zeros(n, bool)
    CH = 1_000_000                                                     # chunked: the index matrix is 8 bytes per base
    for s0 in range(0, n, CH):
        sl = slice(s0, min(n, s0 + CH)); k = sl.stop - sl.start
        idx = pos[sl, None] + np.arange(L)[None, :]
        bx = ACGT[(pac[idx >> 2] >> ((~idx & 3) << 1).astype(np.uint8)) & 3]
        bx = np.where(strand[sl, None] == 1, COMP[bx[:, ::-1]], bx)
        sub = rng.random((k, L)) < 0.01
        is_exact[sl] = ~sub.any(1)
        B[sl, :L] = np.where(sub, ACGT[(np.searchsorted(ACGT, bx) + rng.integers(1, 4, (k, L))) % 4], bx)
        Q[sl, :L] = (33 + rng.integers(20, 41, (k, L))).astype(np.uint8)
    Ln = np.full(n, L, np.uint16)

    kw = dict(mer=a.mer, jump=a.jump, max_kmer_hits=a.max_kmer_hits, nw=0 if a.no_nw else 1, mode=a.mode)
    p = g.Params(**kw)
    from reflib import OracleLib
    orc = OracleLib()
    oix = orc.index_load(fa)
    op = orc.params(**kw)
    pick = rng.integers(0, n, a.sample)
    blk = a.batch if a.batch > 0 else n
    found = checked = bad = mapped = 0
    max_pos = 0
    t_or = 0.0
    batch = g.Batch(ix, min(blk, n), stride)
    for s0 in range(0, n, blk):
        s1 = min(n, s0 + blk)
        log(f"[scale] gm_map_batch reads {s0}..{s1} ...")
        res = batch.map(p, B[s0:s1], Q[s0:s1], Ln[s0:s1])          # full host result (hit lists) for the checks
        log("[scale] gm_map_batch done")
        mb = res["match_begin"]
        mapped += int((res["status"] == 0).sum())
        # property: a read whose substitutions left it exact must have its origin among the reported positions (reads that ended as
        # "too many" report nothing, like on the reference, and are not counted)
        P = res["positions"]
        M = res["matches"]
        if len(M):
            max_pos = max(max_pos, int(max(int(P[m["pos_begin"]:m["pos_end"]]["pos"].max()) for m in M[:: max(1, len(M) // 2000)])))
        ex = np.flatnonzero(is_exact[s0:s1])[: max(1, 2000 * (s1 - s0) // n)]
        for i in ex:
            if res["status"][i] != 0:
                continue
            checked += 1
            ms = M[int(mb[i]):int(mb[i + 1])]
            ok = False
            if len(ms):
                lo, hi = int(ms["pos_begin"].min()), int(ms["pos_end"].max())
                seg = P[lo:hi]
                ok = bool(np.any((seg["pos"] == np.uint64(pos[s0 + i])) & (seg["strand"] == strand[s0 + i])))
            found += ok
        # oracle sample
        t0_ = time.time()
        for gi in pick[(pick >= s0) & (pick < s1)]:
            i = int(gi) - s0
            seq = B[gi, :L].tobytes(); qual = Q[gi, :L].tobytes()
            o = orc.map_read(oix, op, orc.pwm(seq, qual), seq)
            ms = M[int(mb[i]):int(mb[i + 1])]
            same = res["status"][i] == o["status"] and res["denominator"][i] == o["denominator"] and res["top_score"][i] == o["top_score"] and len(ms) == len(o["hits"])
            if same:
                for m, hh in zip(ms, o["hits"]):
                    same &= np.float32(m["score"]).view(np.uint32) == np.float32(hh["score"]).view(np.uint32)
                    same &= [(int(q["pos"]), int(q["strand"])) for q in P[m["pos_begin"]:m["pos_end"]]] == [(int(x), int(y)) for x, y in hh["pos"]]
            bad += not same
        t_or += time.time() - t0_
        log("[scale] checks of the block done")
    c = batch.counters()
    out.update(path=batch.path())
    out.update(exact_reads_checked=checked, exact_reads_origin_found=found, mapped=mapped, max_reported_pos=max_pos)
    out.update(oracle_sample=int(a.sample), oracle_mismatches=int(bad), oracle_reads_per_s=round(a.sample / max(1e-9, t_or), 1),
               oracle_tail_reads=int((pos[pick] >= (1 << 31)).sum()))

    if a.check_output:
        out.update(check_output(g, ix, p, orc, oix, op, B, Q, Ln, L, pick[:a.check_output]))

    # rate with the reads resident in HBM (block by block when --batch is set)
    batch.set_profiling(False)
    dt = 0.0
    ktot = {}
    sa_hits = cands = retries = 0
    for s0 in range(0, n, blk):
        s1 = min(n, s0 + blk)
        batch.upload(p, B[s0:s1], Q[s0:s1], Ln[s0:s1])
        batch.map_device(p)
        batch.counters()                               # reads the device counters back: waits for the launch stream
        batch.kernel_times(); batch.set_profiling(True)
        t0 = time.perf_counter()
        for _ in range(a.steps):
            batch.map_device(p)
        cc = batch.counters()
        dt += (time.perf_counter() - t0) / a.steps
        for k_, (ms_, cnt_) in batch.kernel_times().items():
            if cnt_:
                ktot[k_] = ktot.get(k_, 0.0) + ms_ / a.steps
        batch.set_profiling(False)
        sa_hits += cc["sa_hits"]; cands += cc["candidates"]; retries += cc.get("vote_retries", 0)
    out.update(reads=n, batch=blk, mer=p.mer, max_kmer_hits=p.max_kmer_hits, ms_per_step=round(dt * 1e3, 2), reads_per_s=round(n / dt, 1),
               kernels_ms={k_: round(v_, 3) for k_, v_ in ktot.items()},
               sa_hits_per_read=round(sa_hits / n, 1), candidates_per_read=round(cands / n, 2), vote_retries=retries)
    print(json.dumps(out), flush=True)
    batch.destroy(); ix.close()
    if not a.keep:
        for ext in ("", ".gnumap.pac", ".gnumap.ann", ".gnumap.amb", ".gnumap.bwt", ".gnumap.sa", ".offs.npy"):
            if os.path.exists(fa + ext):
                os.remove(fa + ext)
    return out


if __name__ == "__main__":
    main()
