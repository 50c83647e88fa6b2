"""Parity of the HIP path (through the C ABI of libgnumap_hip.so) with the oracle and the committed reference
vectors.  Bit-exact: SA intervals, located coordinates, fp32 NW score bits, CIGARs, unique-map contents,
fp64 denominators, SAM text.  Only the .sgr track is compared with a tolerance (float atomics; the reference's
own -c > 1 runs are order-dependent in the same way)."""
import ctypes as C
import itertools
import os
import subprocess

import numpy as np
import pytest

import gnumap_amd as g
from reflib import revcomp_pwm, revcomp_str

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def ix_full(syn_fa):
    return g.Index(syn_fa, flags=g.GM_INDEX_FULL_SA)


@pytest.fixture(scope="module")
def ix_sampled(syn_fa):
    return g.Index(syn_fa, flags=0)


@pytest.fixture(scope="module")
def oix(oracle, syn_fa):
    return oracle.index_load(syn_fa)


@pytest.fixture(scope="module")
def packed(syn_reads):
    return g.pack_reads([r[1] for r in syn_reads], [r[2] for r in syn_reads])


def rle(ops: bytes) -> bytes:
    return b"".join(str(len(list(grp))).encode() + bytes([k]) for k, grp in itertools.groupby(ops))


# ------------------------------------------------------------------ building blocks vs the reference's outputs
def test_sa_interval(ix_full, golden):
    kmers = [bytes(k) for k in golden["kmers"]]
    by_len = {}
    for i, k in enumerate(kmers):
        by_len.setdefault(len(k), []).append(i)
    for m, idxs in by_len.items():
        s, e = ix_full.dev_sa_interval([kmers[i] for i in idxs])
        np.testing.assert_array_equal(s, golden["kmer_iv"][idxs, 0])
        np.testing.assert_array_equal(e, golden["kmer_iv"][idxs, 1])


def test_locate_sampled_and_full(ix_full, golden):
    ranks = golden["loc_rank"]
    np.testing.assert_array_equal(ix_full.dev_locate(ranks, False), golden["loc_out"])
    np.testing.assert_array_equal(ix_full.dev_locate(ranks, True), golden["loc_out"])
    # the expanded SA must agree with the LF walk on EVERY rank
    n = ix_full.info.seq_len
    allr = np.arange(1, n + 1, dtype=np.uint64)
    full = ix_full.dev_locate(allr, True)
    np.testing.assert_array_equal(full, ix_full.dev_locate(allr, False))
    assert np.array_equal(np.sort(full), np.arange(n, dtype=np.uint64))        # a permutation of the text positions


def test_nw_score_bits(ix_full, golden, packed):
    B, Q, Ln = packed
    p = g.Params()
    score, valid = ix_full.dev_nw_score(p, B, Q, Ln, golden["nw_read"], golden["nw_rc"].astype(np.uint8), golden["nw_begin"])
    assert valid.all()
    np.testing.assert_array_equal(score.view(np.uint32), golden["nw_score"].view(np.uint32))


def test_window_validity_flags(ix_full, golden, packed):
    B, Q, Ln = packed
    p = g.Params()
    idx = np.array([i for i, L in enumerate(Ln) if L == 100][:1], np.uint32)
    begins = golden["win_begin"][golden["win_len"] == 100]
    expect = np.array([len(bytes(w)) > 0 for w in golden["win_out"][golden["win_len"] == 100]])
    _, valid = ix_full.dev_nw_score(p, B, Q, Ln, np.repeat(idx, len(begins)), np.zeros(len(begins), np.uint8), begins)
    np.testing.assert_array_equal(valid.astype(bool), expect)
    assert (~expect).sum() >= 2


@pytest.mark.parametrize("form", ["table", "direct"])
def test_traceback_cigars(ix_full, golden, packed, form):
    """k_traceback_lane with the substitution values from its LDS table (default) and computed per row (GM_TRACEBACK=direct, also the form
    of rows too long to leave room for the table) against the reference's CIGARs"""
    B, Q, Ln = packed
    p = g.Params()
    g.set_option("GM_TRACEBACK", "direct" if form == "direct" else None)
    try:
        ops = ix_full.dev_traceback(p, B, Q, Ln, golden["nw_read"], golden["nw_rc"].astype(np.uint8), golden["nw_begin"])
    finally:
        g.set_option("GM_TRACEBACK", None)
    for i, o in enumerate(ops):
        assert len(o) == golden["tb_len"][i]
        assert rle(o) == bytes(golden["tb_cigar"][i]), i


def _indel_reads(syn_fa, n, seed, lens=(100, 36, 151, 7, 300)):
    """reads cut from chrA with substitutions AND short insertions / deletions (1 .. 6 bases), so that the best path uses the band"""
    rng = np.random.default_rng(seed)
    genome = b"".join(l.strip() for l in open(syn_fa, "rb") if not l.startswith(b">")).upper().replace(b"N", b"A")
    reads, begins = [], []
    for k in range(n):
        L = int(lens[k % len(lens)])
        p0 = int(rng.integers(1000, 130_000))
        s = bytearray(genome[p0:p0 + L + 16])
        for _ in range(int(rng.integers(0, 4))):
            q = int(rng.integers(1, max(2, L - 8))); w = int(rng.integers(1, 7))
            if rng.integers(0, 2): del s[q:q + w]
            else: s[q:q] = bytes(b"ACGT"[int(x)] for x in rng.integers(0, 4, w))
        s = s[:L]
        for q in rng.integers(0, L, max(1, L // 30)):
            s[q] = b"ACGT"[int(rng.integers(0, 4))]
        reads.append((f"indel{k}_{p0}", bytes(s), bytes((33 + rng.integers(2, 41, L)).astype(np.uint8))))
        begins.append(p0)
    return reads, begins


@pytest.mark.parametrize("G", [1, 2, 3, 4, 5, 7])
def test_band_width_kernels_match_oracle(G, ix_full, oracle, oix, syn_fa):
    """-M / --max_gap: DP score bits and traceback operations of the generic band kernels (gm_band.hip; G = 3 runs the register-band
    kernels on the same inputs) against the oracle's restatement of bin_seq.cpp:781-850 / :445-718, candidates shifted by up to
    G + 2 bases either way so that paths run along and into the band's edge"""
    reads, begins = _indel_reads(syn_fa, 60, 100 + G)
    B, Q, Ln = g.pack_reads([r[1] for r in reads], [r[2] for r in reads])
    p = g.Params(max_gap=G); op = oracle.params(max_gap=G)
    ridx, strand, pos = [], [], []
    for i, b0 in enumerate(begins):
        for sh in (0, -1, 2, -(G + 2), G + 1):
            for st in (0, 1):
                ridx.append(i); strand.append(st); pos.append(b0 + sh)
    ridx = np.array(ridx, np.uint32); strand = np.array(strand, np.uint8); pos = np.array(pos, np.uint64)
    score, valid = ix_full.dev_nw_score(p, B, Q, Ln, ridx, strand, pos)
    ops = ix_full.dev_traceback(p, B, Q, Ln, ridx, strand, pos)
    assert valid.all()
    n_gapped = 0
    for k in range(len(ridx)):
        name, seq, qual = reads[ridx[k]]
        P = oracle.pwm(seq, qual); cons = seq
        if strand[k]:
            P = revcomp_pwm(P); cons = revcomp_str(seq).upper()
        w = oracle.window(oix, int(pos[k]), len(seq))
        assert len(w) == len(seq)
        want = np.float32(oracle.lib.gmo_nw_score(C.byref(op), np.ascontiguousarray(P, np.float32), len(seq), w))
        assert score[k].view(np.uint32) == want.view(np.uint32), (name, k, score[k], want)
        _, alen, cig = oracle.traceback(op, P, cons, w)
        assert len(ops[k]) == alen and rle(ops[k]) == cig, (name, k, rle(ops[k]), cig)
        n_gapped += (b"I" in cig) or (b"D" in cig)
    assert n_gapped > 50


# ------------------------------------------------------------------ the whole hot path vs the oracle
CONFIGS = {
    "default": {},
    "no_nw": dict(nw=0),
    "m16": dict(mer=16),
    "m12_j3": dict(mer=12, jump=3),
    "h30": dict(max_kmer_hits=30),
    "T2": dict(max_matches=2),
    "unique": dict(unique_only=1),
    "unique_no_nw": dict(unique_only=1, nw=0),
    "k1": dict(min_seed_hits=1, mer=14),
    "k3": dict(min_seed_hits=3),
    "up": dict(neg_strand=0),
    "down": dict(pos_strand=0),
    "bs": dict(mode=1),
    "atog": dict(mode=3),
    "fast": dict(fast=1, mer=14, jump=14),                     # one seed only -> never reaches -k 2 votes (same in the reference)
    "fast_k1": dict(fast=1, mer=14, jump=14, min_seed_hits=1),
    "raw60": dict(align_score=60.0, align_is_fraction=0),
    "a07_q50": dict(align_score=0.7, cutoff=50.0),
    "m20_j2": dict(mer=20, jump=2),                             # 40 seeds per strand: 64-bit step masks inside k_vote_slots (<= 40 slots)
    "m6_j2": dict(mer=6, jump=2),                               # 48 seeds x ~70 hits per strand: > 32 seeds (64-bit step masks), > 40 slots (list kernel)
    "M1": dict(max_gap=1),                                      # -M: generic band kernels (gm_band.hip)
    "M2_a07": dict(max_gap=2, align_score=0.7),
    "M5_bs": dict(max_gap=5, mode=1),
    "M7_k1": dict(max_gap=7, min_seed_hits=1, mer=14),
}


def _oracle_results(oracle, oix, op, syn_reads):
    out = []
    for name, seq, qual in syn_reads:
        P = oracle.pwm(seq, qual) if len(seq) else np.zeros((1, 4), np.float32)
        out.append(oracle.map_read(oix, op, P, seq))
    return out


def _compare(res, ores, syn_reads):
    mb = res["match_begin"]
    for i, o in enumerate(ores):
        name = syn_reads[i][0]
        assert res["status"][i] == o["status"], (name, res["status"][i], o["status"])
        if o["status"] in (0, 1, 2):
            assert np.float32(res["self_score"][i]).view(np.uint32) == np.float32(o["self_score"]).view(np.uint32), name
        assert res["top_score"][i] == o["top_score"], (name, res["top_score"][i], o["top_score"])
        assert res["denominator"][i] == o["denominator"], (name, res["denominator"][i], o["denominator"])
        ms = res["matches"][int(mb[i]):int(mb[i + 1])]
        assert len(ms) == len(o["hits"]), name
        for m, h in zip(ms, o["hits"]):                       # std::map (key) order on both sides
            assert np.float32(m["score"]).view(np.uint32) == np.float32(h["score"]).view(np.uint32), name
            assert m["first_strand"] == h["first_strand"], name
            pos = [(int(q["pos"]), int(q["strand"])) for q in res["positions"][m["pos_begin"]:m["pos_end"]]]
            assert pos == [(int(a), int(b)) for a, b in h["pos"]], name
            assert (int(m["first_pos"]), int(m["first_strand"])) in pos


@pytest.mark.parametrize("cfg", list(CONFIGS))
def test_map_batch_matches_oracle(cfg, ix_full, oracle, oix, syn_reads, packed):
    kw = CONFIGS[cfg]
    p = g.Params(**kw)
    op = oracle.params(**kw)
    B, Q, Ln = packed
    batch = g.Batch(ix_full, len(syn_reads), B.shape[1])
    res = batch.map(p, B, Q, Ln)
    ores = _oracle_results(oracle, oix, op, syn_reads)
    _compare(res, ores, syn_reads)
    assert sum(o["status"] == 0 for o in ores) > (-1 if cfg in ("a07_q50", "fast") else 50)
    batch.destroy()


def test_sampled_locate_path_equals_full_sa_path(ix_full, ix_sampled, syn_reads, packed):
    B, Q, Ln = packed
    p = g.Params()
    out = []
    for ix in (ix_full, ix_sampled):
        batch = g.Batch(ix, len(syn_reads), B.shape[1])
        batch.upload(p, B, Q, Ln)
        batch.map_device(p)
        hits, status, self_score, top = batch.raw_hits()
        out.append((hits.tobytes(), status.tobytes(), self_score.tobytes(), top.tobytes(), batch.counters()))
        batch.destroy()
    assert out[0][:4] == out[1][:4]
    assert out[1][4]["lf_steps"] > 10 * out[1][4]["sa_hits"] and out[0][4]["lf_steps"] == 0


def test_counters_match_oracle_work(ix_sampled, oracle, oix, syn_reads, packed):
    """the kernel-side work counters (algorithmic-bytes accounting) against the oracle's own counts"""
    B, Q, Ln = packed
    p = g.Params()
    batch = g.Batch(ix_sampled, len(syn_reads), B.shape[1])
    batch.upload(p, B, Q, Ln)
    batch.map_device(p)
    c = batch.counters()
    ores = _oracle_results(oracle, oix, oracle.params(), syn_reads)
    assert c["sa_hits"] == sum(o["ctr"]["locates"] for o in ores)
    assert c["lf_steps"] == sum(o["ctr"]["lf_steps"] for o in ores)
    assert c["candidates"] >= sum(o["ctr"]["nw"] for o in ores)          # the kernel also lists windows that fail the contig test
    assert c["kmers_searched"] <= sum(o["ctr"]["kmers"] for o in ores)   # failed k-mers are skipped in one jump on the device
    batch.destroy()


def test_small_and_ragged_batches(ix_full, oracle, oix, syn_reads):
    p = g.Params(); op = oracle.params()
    for sel in ([0], [5, 6], list(range(500, len(syn_reads))), [len(syn_reads) - 1]):
        reads = [syn_reads[i] for i in sel]
        B, Q, Ln = g.pack_reads([r[1] for r in reads], [r[2] for r in reads])
        batch = g.Batch(ix_full, len(reads), B.shape[1])
        res = batch.map(p, B, Q, Ln)
        _compare(res, _oracle_results(oracle, oix, op, reads), reads)
        batch.destroy()
    batch = g.Batch(ix_full, 4, 8)
    res = batch.map(p, np.zeros((0, 8), np.uint8), np.zeros((0, 8), np.uint8), np.zeros(0, np.uint16))
    assert len(res["matches"]) == 0


def _long_reads(syn_fa, L, n, seed):
    rng = np.random.default_rng(seed)
    genome = b"".join(l.strip() for l in open(syn_fa, "rb") if not l.startswith(b">")).upper().replace(b"N", b"A")
    reads = []
    for k in range(n):
        p0 = int(rng.integers(0, 140_000 - L))           # inside chrA
        s = bytearray(genome[p0:p0 + L])
        for q in rng.integers(0, L, max(1, L // 80)):
            s[q] = b"ACGT"[int(rng.integers(0, 4))]
        if k % 2:
            s = bytearray(revcomp_str(bytes(s)).upper())
        reads.append((f"long{k}_{p0}", bytes(s), bytes((33 + rng.integers(20, 41, L)).astype(np.uint8))))
    return reads


@pytest.mark.parametrize("L,kw", [(250, {}), (330, dict(mer=12, jump=4)), (600, {}), (1000, dict(mer=16, jump=8)),
                                  (1500, dict(mer=16, jump=8)), (2048, dict(mer=20, jump=10)), (600, dict(max_gap=5)), (2048, dict(mer=20, jump=10, max_gap=1))])      # > 1280 bases: k_seed tiles of fewer reads (LDS budget)
def test_long_reads_all_vote_kernels(L, kw, ix_full, oracle, oix, syn_fa):
    """L=250: 64-bit step masks; L=600 at j=5: more than 64 seeds -> the ordered vote kernel"""
    reads = _long_reads(syn_fa, L, 24, L)
    B, Q, Ln = g.pack_reads([r[1] for r in reads], [r[2] for r in reads])
    p = g.Params(**kw); op = oracle.params(**kw)
    batch = g.Batch(ix_full, len(reads), B.shape[1])
    res = batch.map(p, B, Q, Ln)
    ores = _oracle_results(oracle, oix, op, reads)
    _compare(res, ores, reads)
    assert sum(o["status"] == 0 for o in ores) >= 20
    recs, cigars = batch.output(p, res)
    assert len(recs) >= 20
    batch.destroy()


def test_repeat_heavy_reads_take_the_retry_path(ix_full, oracle, oix, syn_fa):
    """low-complexity reads collect thousands of multi-voted positions: the LDS vote table overflows and the
    global-table kernel takes over; results must not change"""
    unit = b"ACACACGT"
    reads = [(f"lc{k}", (unit * 20)[k:k + 100], b"I" * 100) for k in range(8)]
    B, Q, Ln = g.pack_reads([r[1] for r in reads], [r[2] for r in reads])
    for kw in (dict(mer=6, jump=1, max_matches=100000), dict(mer=6, jump=1, nw=0)):
        p = g.Params(**kw); op = oracle.params(**kw)
        batch = g.Batch(ix_full, len(reads), B.shape[1])
        res = batch.map(p, B, Q, Ln)
        _compare(res, _oracle_results(oracle, oix, op, reads), reads)
        assert batch.counters()["vote_retries"] > 0
        batch.destroy()


def test_bad_quality_is_an_error(ix_full):
    p = g.Params()
    B, Q, Ln = g.pack_reads([b"ACGTACGTACGTACGTACGT"], [b"IIIIIIIIII IIIIIIIII"])      # ' ' = Phred -1
    batch = g.Batch(ix_full, 1, B.shape[1])
    with pytest.raises(g.GnumapError, match="Invalid Fastq Character"):
        batch.map(p, B, Q, Ln)


def test_illumina_fallback(ix_full, oracle, oix, syn_reads):
    """--illumina (Phred+64) with the automatic switch-off at the first read that shows a quality below '@'"""
    reads = []
    for i, (n, s, q) in enumerate(syn_reads[:12]):
        if i < 5:
            q = bytes(min(126, c + 31) for c in q)            # Phred+64 encoded
        reads.append((n, s, q))
    B, Q, Ln = g.pack_reads([r[1] for r in reads], [r[2] for r in reads])
    p = g.Params(illumina=1)
    batch = g.Batch(ix_full, len(reads), B.shape[1])
    res = batch.map(p, B, Q, Ln)
    op64 = oracle.params(illumina=1); op33 = oracle.params()
    ores = []
    for i, (n, s, q) in enumerate(reads):
        P = oracle.pwm(s, q, illumina=1 if i < 5 else 0)
        ores.append(oracle.map_read(oix, op64 if i < 5 else op33, P, s))
    _compare(res, ores, reads)


# ------------------------------------------------------------------ end to end: the gnumap binary vs the oracle's run
def _run_cli(args, out_prefix, syn_fa, syn_fq):
    exe = os.path.join(ROOT, "gnumap_amd", "bin", "gnumap")
    r = subprocess.run([exe, "-g", syn_fa, "-o", out_prefix] + args + [syn_fq], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr
    return r


def _sgr(path):
    d = {}
    for line in open(path):
        c, p, v = line.split("\t")
        d[(c, int(p))] = float(v)
    return d


def _gmp(path):
    d = {}
    for line in open(path):
        f = line.rstrip("\n").split("\t")
        assert len(f) == 8, line
        d[(f[0], int(f[1]))] = [float(x) for x in f[2:]]
    return d


@pytest.mark.parametrize("name,args,kw", [
    ("default", [], {}),
    ("no_nw", ["--no_nw"], dict(nw=0)),
    ("m16_h150_all", ["-m", "16", "-h", "150", "--print_all_sam"], dict(mer=16, max_kmer_hits=150, print_all_sam=1)),
    ("bs", ["-b"], dict(mode=1)),
    ("b2", ["--b2"], dict(mode=2)),
    ("a_to_g", ["-d"], dict(mode=3)),
    ("sampled", ["--locate=sampled"], {}),
    ("batch64", ["--batch=64"], {}),
])
def test_cli_sam_identical_to_oracle(name, args, kw, tmp_path, oracle, oix, syn_fa, syn_fq):
    mine = str(tmp_path / "mine"); ref = str(tmp_path / "orc")
    _run_cli(args, mine, syn_fa, syn_fq)
    oracle.run(oix, oracle.params(**kw), syn_fq, ref, threads=4)
    a = [l for l in open(mine + ".sam") if not l.startswith("@PG")]
    b = [l for l in open(ref + ".sam") if not l.startswith("@PG")]
    assert len(a) > 400
    assert a == b
    if kw.get("mode", 0):
        # -b / -d write <out>.gmp (GenomeBwt::PrintFinalBisulfite) instead of <out>.sgr: name, pos, amount, a c g t n
        assert not os.path.exists(mine + ".sgr") and not os.path.exists(ref + ".sgr")
        ga, gb = _gmp(mine + ".gmp"), _gmp(ref + ".gmp")
        assert len(gb) > 1000
        # a locus whose fp32-atomic sum is a rounding error away from 0 may differ in presence only
        assert all(max(ga.get(k, gb.get(k))[:1]) < 2e-3 for k in set(ga) ^ set(gb))
        for k in set(ga) & set(gb):
            for x, y in zip(ga[k], gb[k]):
                assert abs(x - y) <= 1e-4 * max(1.0, abs(y)) + 2e-5, k
        return
    sa, sb = _sgr(mine + ".sgr"), _sgr(ref + ".sgr")
    assert set(sa) == set(sb) or max(abs(sa.get(k, 0) - sb.get(k, 0)) for k in set(sa) | set(sb)) < 2e-3
    for k in sb:
        assert abs(sa.get(k, 0.0) - sb[k]) <= 1e-4 * max(1.0, abs(sb[k])) + 2e-5, k


def test_coverage_device_view_for_rccl(ix_full):
    """bench.py all-reduces the HBM-resident coverage track in place through a zero-copy torch view"""
    import torch
    from gnumap_amd import dist as gd
    ix_full.coverage_reset(8)
    t = gd.DeviceTrack(ix_full.coverage_device_ptr(), ix_full.coverage_bins()).tensor(torch.device("cuda", 0))
    assert t.dtype == torch.float32 and t.numel() == ix_full.coverage_bins() and float(t.abs().sum()) == 0.0
    t[5] = 2.5; t[-1] = 1.0
    gd.allreduce_coverage(t)            # world size 1: identity
    torch.cuda.synchronize()
    host = ix_full.coverage_download()
    assert host[5] == 2.5 and host[-1] == 1.0 and host.sum() == 3.5
    ix_full.coverage_reset(8)


def test_rccl_allreduce_path_on_one_rank(ix_full, monkeypatch):
    """gm_coverage_allreduce (what `gnumap --gpus N` calls): with GM_RCCL_FORCE=1 a single GPU goes through ncclCommInitAll /
    ncclAllReduce (in place, own stream) / async-error check / destroy; a one-rank sum must leave the track as it was"""
    monkeypatch.setenv("GM_RCCL_FORCE", "1")
    ix_full.coverage_reset(8)
    ix_full.coverage_add([800, 5000, 5004], [64, 16, 8], [1.5, 2.0, 0.25])
    before = ix_full.coverage_download()
    assert before.sum() == 64 * 1.5 + 16 * 2.0 + 8 * 0.25 and before[100] == 12.0
    arr = (C.c_void_p * 1)(ix_full.h)
    assert g.lib().gm_coverage_allreduce(arr, 1) == 0, g.lib().gm_last_error()
    np.testing.assert_array_equal(ix_full.coverage_download(), before)
    monkeypatch.delenv("GM_RCCL_FORCE")
    assert g.lib().gm_coverage_allreduce(arr, 1) == 0
    ix_full.coverage_reset(8)


def test_batch_limits_are_refused_loudly(ix_full):
    for reads, ln in ((16_000_001, 100), (1000, 2049)):
        with pytest.raises(g.GnumapError):
            g.Batch(ix_full, reads, ln)


def test_pipelined_sub_batches_equal_single_pass(ix_full, syn_reads, packed, monkeypatch):
    """GM_PIPELINE=<n>: sub-batches over three streams must give the same raw hits as the single pass"""
    B, Q, Ln = packed
    p = g.Params()
    reps = 40                                   # 551 x 40 reads so that several sub-batches of 4096 exist
    Bb = np.tile(B, (reps, 1)); Qb = np.tile(Q, (reps, 1)); Lb = np.tile(Ln, reps)
    out = []
    for env in (None, "4096"):
        if env is None:
            monkeypatch.delenv("GM_PIPELINE", raising=False)
        else:
            monkeypatch.setenv("GM_PIPELINE", env)
        batch = g.Batch(ix_full, len(Lb), Bb.shape[1])
        batch.upload(p, Bb, Qb, Lb)
        batch.map_device(p)
        hits, status, self_score, top = batch.raw_hits()
        out.append((hits.tobytes(), status.tobytes(), self_score.tobytes(), top.tobytes(), batch.counters()))
        batch.destroy()
    assert out[0][:4] == out[1][:4]
    assert out[0][4]["sa_hits"] == out[1][4]["sa_hits"] and out[0][4]["candidates"] == out[1][4]["candidates"]
    n = len(syn_reads)
    h = np.frombuffer(out[1][0], dtype=g.api.RAW_HIT_DTYPE)
    first = h[h["read"] < n]; last = h[h["read"] >= n * (reps - 1)]
    assert len(first) == len(last) and np.array_equal(first["pos"], last["pos"])


_VARIANT_ORACLE = {}


VARIANT_ENVS = [dict(GM_VOTE="block"),                                             # dense seeds: k_vote_slots (the default dense kernel)
                                 dict(GM_VOTE="big"), dict(GM_VOTE="rounds"),                      # its 64-slot form; rounds of the block form
                                 dict(GM_VOTE="block", GM_VOTE_SLOTS="10"), dict(GM_VOTE="block", GM_VOTE_SLOTS="20"), dict(GM_VOTE="block", GM_VOTE_SLOTS="40"),   # 16- / 24- / 40-slot forms (+ list kernel)
                                 dict(GM_VOTE="block", GM_VOTE_SLOTS="0"), dict(GM_VOTE="block", GM_VOTE_SLOTS="0", GM_TEST_SAMPLED="1"),   # k_vote_tiny (+ list kernel, retry kernel)
                                 dict(GM_VOTE="block", GM_VOTE_KERNEL="block"), dict(GM_VOTE="block", GM_VOTE_KERNEL="block", GM_VOTE_NT="64"),
                                 dict(GM_VOTE="block", GM_VOTE_KERNEL="block", GM_VOTE_NT="256"), dict(GM_VOTE="block", GM_VOTE_KERNEL="block", GM_VOTE_TB="10"),
                                 dict(GM_VOTE="block", GM_VOTE_KERNEL="block", GM_RETRY_BUDGET="16384"),     # retry tables handed out in several launches
                                 dict(GM_VOTE="block", GM_TEST_SAMPLED="1"),                         # k_vote_slots on LF-walk coordinates (no full SA)
                                 dict(GM_VOTE="wave"), dict(GM_VOTE="wave", GM_VOTE_SPARSE="0"),
                                 dict(GM_NW="wave"), dict(GM_KMER_TABLE="0"), dict(GM_KMER_TABLE="6"), dict(GM_KMER_COMPACT="0"),
                                 dict(GM_KMER_TABLE="13"), dict(GM_KMER_TABLE="14"),           # tables extended one character at a time (m20_j2, k1)
                                 dict(GM_KMER_TABLE="15"), dict(GM_KMER_TABLE="16"),           # 2^30 / 2^32 codes (64-bit code arithmetic), 43 GB of HBM at 16
                                 dict(GM_VOTE="block", GM_VOTE_SLOTS="-1"), dict(GM_VOTE="block", GM_VOTE_SLOTS="-1", GM_TEST_SAMPLED="1"),   # k_vote_tiny2
                                 dict(GM_VOTE="block", GM_VOTE_SLOTS="0", GM_VOTE_FIXED="0"), dict(GM_VOTE="block", GM_VOTE_SLOTS="-1", GM_VOTE_FIXED="0"),   # candidates through the bump counters instead of own slots + k_cand_gather
                                 dict(GM_NW_ROWS="0"), dict(GM_PREP="tile"),                          # k_nw_lane without the row registers; k_prep (LDS tiles) instead of k_prep_rows
                                 dict(GM_HEAVY_MIN="64"), dict(GM_HEAVY_MIN="8", GM_HEAVY_BUDGET="200000"),      # sorted-key path for read x strands with many SA hits (several chunks)
                                 dict(GM_HEAVY_MIN="64", GM_TEST_SAMPLED="1"),
                                 # the one-wave kernels look their seeds up themselves on the full SA (fused form: chosen when every k-mer is expected
                                 # >= 8 times in the reference, forced here with GM_SEED_FUSED=1 - on this small reference most reads then take the
                                 # serial walk of the irregular cases, which is the point): the two-kernel form, a table without its compact form
                                 # (falls back to k_seed), 14- / 16-mers with a table that long, hand-over to the heavy path
                                 dict(GM_VOTE="block", GM_VOTE_SLOTS="0", GM_SEED_FUSED="0"), dict(GM_VOTE="block", GM_VOTE_SLOTS="-1", GM_SEED_FUSED="0"),
                                 dict(GM_VOTE="block", GM_VOTE_SLOTS="0", GM_SEED_FUSED="1"), dict(GM_VOTE="block", GM_VOTE_SLOTS="-1", GM_SEED_FUSED="1"),
                                 dict(GM_VOTE="block", GM_VOTE_SLOTS="0", GM_SEED_FUSED="1", GM_KMER_COMPACT="0"), dict(GM_VOTE="block", GM_VOTE_SLOTS="0", GM_SEED_FUSED="1", GM_KMER_TABLE="16"),
                                 dict(GM_VOTE="block", GM_VOTE_SLOTS="-1", GM_SEED_FUSED="1", GM_KMER_TABLE="14"),
                                 dict(GM_VOTE="block", GM_VOTE_SLOTS="0", GM_SEED_FUSED="1", GM_HEAVY_MIN="64"),
                                 dict(GM_VOTE="block", GM_VOTE_SLOTS="-1", GM_SEED_FUSED="1", GM_HEAVY_MIN="8", GM_HEAVY_BUDGET="200000"),
                                 # the same inside k_vote_slots (wave 0 of the workgroup looks the seeds up): every slot form, hand-over to the list / heavy kernels
                                 dict(GM_VOTE="block", GM_SEED_FUSED="1"), dict(GM_VOTE="big", GM_SEED_FUSED="1"), dict(GM_VOTE="block", GM_VOTE_SLOTS="10", GM_SEED_FUSED="1"),
                                 dict(GM_VOTE="block", GM_VOTE_SLOTS="20", GM_SEED_FUSED="1"), dict(GM_VOTE="block", GM_SEED_FUSED="1", GM_HEAVY_MIN="64"),
                                 # the 64-slot form of those runs k_vote_slots_pp by default (persistent workgroups, the next read x strands' loads in
                                 # flight), GM_SLOTS_PIPE=1 puts every form on it; here: chunks of 2, of an odd number and one chunk for everything
                                 # (pipeline fill and drain, a chunk that ends inside a read), the hand-over to the heavy path from inside the pipeline,
                                 # and the one-workgroup-per-read-x-strand kernel of the 64-slot form (GM_SLOTS_PIPE=0)
                                 dict(GM_VOTE="block", GM_SEED_FUSED="1", GM_SLOTS_PIPE="1"), dict(GM_VOTE="block", GM_SEED_FUSED="1", GM_SLOTS_PIPE="1", GM_SLOTS_CHUNK="2"),
                                 dict(GM_VOTE="big", GM_SEED_FUSED="1", GM_SLOTS_CHUNK="7"),
                                 dict(GM_VOTE="block", GM_VOTE_SLOTS="10", GM_SEED_FUSED="1", GM_SLOTS_PIPE="1", GM_SLOTS_CHUNK="100000"),
                                 dict(GM_VOTE="block", GM_VOTE_SLOTS="20", GM_SEED_FUSED="1", GM_SLOTS_PIPE="1"),
                                 dict(GM_VOTE="block", GM_SEED_FUSED="1", GM_HEAVY_MIN="64", GM_SLOTS_PIPE="1"),
                                 dict(GM_VOTE="big", GM_SEED_FUSED="1", GM_SLOTS_PIPE="0"),
                                 # ... and its form that takes k_seed's seed rows (what GM_VOTE=big alone runs by now)
                                 dict(GM_VOTE="big", GM_SLOTS_PIPE="0"), dict(GM_VOTE="big", GM_SLOTS_CHUNK="3"), dict(GM_VOTE="block", GM_SLOTS_PIPE="1"),
                                 dict(GM_VOTE="block", GM_VOTE_SLOTS="10", GM_SLOTS_PIPE="1"), dict(GM_VOTE="block", GM_VOTE_SLOTS="20", GM_SLOTS_PIPE="1", GM_SLOTS_CHUNK="100000")]
VARIANT_CFGS = ["default", "no_nw", "k3", "h30", "m6_j2", "m20_j2", "k1"]


def _variant_skip(env, cfg):
    # the whole grid is 50 x 7 processes; the non-default forms run on the configurations that reach their special cases
    if "GM_SEED_FUSED" in env and cfg in ("no_nw", "k3", "m20_j2"):
        return "fused seed lookup: covered by default / h30 / m6_j2 / k1 (m20_j2 is never fused: the table is shorter than the seed)"
    if env.get("GM_VOTE_KERNEL") == "block" and cfg in ("no_nw", "h30", "k1", "m20_j2"):
        return "block form of the dense kernel (not the default): default / k3 / m6_j2"
    if env.get("GM_KMER_TABLE") in ("13", "15") and cfg in ("no_nw", "k3", "h30"):
        return "odd table lengths matter for the long seeds"
    return None


@pytest.mark.parametrize("env", VARIANT_ENVS,
                         ids=lambda e: ",".join(f"{k[3:]}={v}" for k, v in e.items()))
@pytest.mark.parametrize("cfg", VARIANT_CFGS)
def test_every_kernel_variant_matches_oracle(env, cfg, syn_fa, oracle, oix, syn_reads, packed, tmp_path):
    """the dispatch heuristics pick kernels by seed density; force each variant on the same inputs and compare the result of
    gm_map_batch (status, self / top score, denominator, matches in key order, position sets) with the ORACLE, read by read"""
    why = _variant_skip(env, cfg)
    if why:
        pytest.skip(why)
    _prefetch_variants(syn_fa)                        # the grid's processes, three at a time, started by the first case that gets here
    _run_variant(env, cfg, syn_fa, oracle, oix, syn_reads, tmp_path)


@pytest.mark.parametrize("cfg", ["default", "no_nw", "T2", "unique", "unique_no_nw", "k1", "m6_j2", "bs", "a07_q50"])
def test_big_grouping_path_matches_oracle(cfg, syn_fa, oracle, oix, syn_reads, tmp_path):
    """GM_GROUP_BIG_MIN=1 sends every read with >= 2 accepted hits through the hash-set + wave-radix-sort grouping (k_group_big /
    k_group_write_big: the path of reads with thousands of repeat copies), incl. its -T / -u exits and the hand-back cases"""
    _run_variant(dict(GM_GROUP_BIG_MIN="1"), cfg, syn_fa, oracle, oix, syn_reads, tmp_path)


_VARIANT_JOBS = {}
_VARIANT_POOL = []


def _variant_key(env, cfg):
    return (tuple(sorted(env.items())), cfg)


def _variant_process(env, cfg, syn_fa, out):
    import subprocess, sys
    # kernel-variant switches are cached in static locals on first use, so each combination runs in its own process
    code = f"""
import sys, os, numpy as np
sys.path.insert(0, {ROOT!r}); sys.path.insert(0, {os.path.join(ROOT, 'tests')!r})
import gnumap_amd as g
from conftest import read_fastq
reads = read_fastq({os.path.join(ROOT, 'tests', 'golden', 'syn.fq')!r})
B, Q, Ln = g.pack_reads([r[1] for r in reads], [r[2] for r in reads])
ix = g.Index({syn_fa!r}, flags=0 if os.environ.get('GM_TEST_SAMPLED') else g.GM_INDEX_FULL_SA)
p = g.Params(**{CONFIGS[cfg]!r})
b = g.Batch(ix, len(reads), B.shape[1])
res = b.map(p, B, Q, Ln)
ctr = b.counters()
np.savez({out!r}, **{{k: v for k, v in res.items() if not k.startswith('_')}}, **{{'ctr_' + k: np.int64(v) for k, v in ctr.items()}})
"""
    return subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=dict(os.environ, **env), timeout=600)


def _prefetch_variants(syn_fa):
    """the ~220 processes of test_every_kernel_variant_matches_oracle (1.5 s each, mostly start-up) run three at a time in the background -
    with the pytest process that is 4 on the card, under the box's limit of 6; a case waits for its own process only"""
    if _VARIANT_POOL:
        return
    import tempfile
    from concurrent.futures import ThreadPoolExecutor
    pool = ThreadPoolExecutor(max_workers=3)
    _VARIANT_POOL.append(pool)
    import atexit
    atexit.register(lambda: pool.shutdown(wait=True, cancel_futures=True))       # a run that stops early does not start the rest of the grid
    d = tempfile.mkdtemp(prefix="gm_variants_")
    k = 0
    for cfg in VARIANT_CFGS:                          # pytest's order for the stacked parametrisations: the one next to the function varies slowest
        for env in VARIANT_ENVS:
            if _variant_skip(env, cfg):
                continue
            out = os.path.join(d, f"v{k}.npz"); k += 1
            _VARIANT_JOBS[_variant_key(env, cfg)] = (pool.submit(_variant_process, env, cfg, syn_fa, out), out)


def _run_variant(env, cfg, syn_fa, oracle, oix, syn_reads, tmp_path):
    job = _VARIANT_JOBS.pop(_variant_key(env, cfg), None)
    if job is not None:
        r, out = job[0].result(), job[1]
    else:
        out = str(tmp_path / "res.npz")
        r = _variant_process(env, cfg, syn_fa, out)
    assert r.returncode == 0, r.stderr[-1500:]
    res = dict(np.load(out))
    if cfg not in _VARIANT_ORACLE:
        _VARIANT_ORACLE[cfg] = _oracle_results(oracle, oix, oracle.params(**CONFIGS[cfg]), syn_reads)
    _compare(res, _VARIANT_ORACLE[cfg], syn_reads)
    assert len(res["matches"]) > 400
    return {k[4:]: int(v) for k, v in res.items() if k.startswith("ctr_")}


@pytest.mark.parametrize("cfg", ["default", "h30", "m6_j2"])
@pytest.mark.parametrize("slots", ["0", "-1", "40"])
def test_fused_seed_lookup_counts_the_same_work(cfg, slots, syn_fa, oracle, oix, syn_reads, tmp_path):
    """seed lookup inside k_vote_tiny / k_vote_tiny2 / k_vote_slots (lane j takes the k-mer at j * jump; a failed or capped k-mer or an N sends the
    read x strand through the serial walk) against the k_seed form: same results (both compared with the oracle) and the same work
    counters - k-mers searched, table probes, seeds, SA hits, candidates, DP cells"""
    env = dict(GM_VOTE="block", GM_VOTE_SLOTS=slots, GM_KMER_TABLE="14")
    fused = _run_variant(dict(env, GM_SEED_FUSED="1"), cfg, syn_fa, oracle, oix, syn_reads, tmp_path)
    plain = _run_variant(dict(env, GM_SEED_FUSED="0"), cfg, syn_fa, oracle, oix, syn_reads, tmp_path)
    assert fused == plain, (fused, plain)
    assert fused["seeds_used"] > 5000 and fused["kmers_searched"] >= fused["seeds_used"]
