"""k_vote_bucket (gm_bucket.hip): seed lookup + locate + votes of a whole read per wavefront through the k-mer -> positions records.
The small fixture reference (400 kbp) is the hard case for it: at -m 10 most k-mers of a read with an error do not occur (the
halves walk again round by round), at -m 8 / -m 7 a k-mer occurs 6 / 24 times (inline records, records beyond 31 hits that go
through the suffix array, read x strands handed to the list / heavy kernels).  Every configuration is compared with the ORACLE read
by read (status, self / top score, denominator, matches in key order, position sets), in ONE process: the library's switches are
read on every call (gm_set_option)."""
import numpy as np
import pytest

import gnumap_amd as g
from test_gpu_parity import _compare, _long_reads, _oracle_results

pytestmark = pytest.mark.gpu

BUCKET_CONFIGS = {
    "default": {},                                              # -m 10 -j 5: up to 29 seeds per strand (150-bp reads) -> k_vote_bucket<8>
    "h30": dict(max_kmer_hits=30),
    "k3": dict(min_seed_hits=3),
    "no_nw": dict(nw=0),
    "up": dict(neg_strand=0),
    "down": dict(pos_strand=0),
    "T2": dict(max_matches=2),
    "unique": dict(unique_only=1),
    "m9_j5": dict(mer=9, jump=5),
    "m8_j5": dict(mer=8, jump=5),                               # ~6 hits per k-mer: inline records; some beyond 31 hits
    "m8_j5_h10": dict(mer=8, jump=5, max_kmer_hits=10),         # most seeds capped: the walk slides base by base (:213-217)
    "m8_j5_h31": dict(mer=8, jump=5, max_kmer_hits=31),
    "m8_j5_h32_no_nw": dict(mer=8, jump=5, max_kmer_hits=32, nw=0),
    "m8_j6_k3": dict(mer=8, jump=6, min_seed_hits=3),
    "m7_j5": dict(mer=7, jump=5),                               # ~24 hits per k-mer: many records beyond 31 hits, ~700 hits per strand -> list kernel
    "m7_j5_h40": dict(mer=7, jump=5, max_kmer_hits=40),
    "m7_j9": dict(mer=7, jump=9),                               # 16 seeds per strand at most -> k_vote_bucket<4> .. <6>
    "m12_j6": dict(mer=12, jump=6),
    "m14_j7": dict(mer=14, jump=7),
    "fast": dict(fast=1, mer=14, jump=14),
    "fast_m8": dict(fast=1, mer=8, jump=8, min_seed_hits=2),
    "m12_j18": dict(mer=12, jump=18),                           # 8 seeds per strand at most -> k_vote_bucket<2>
}

@pytest.fixture(scope="module")
def ix_full(syn_fa):
    return g.Index(syn_fa, flags=g.GM_INDEX_FULL_SA)


@pytest.fixture(scope="module")
def oix(oracle, syn_fa):
    return oracle.index_load(syn_fa)


@pytest.fixture(scope="module")
def packed(syn_reads):
    return g.pack_reads([r[1] for r in syn_reads], [r[2] for r in syn_reads])


@pytest.fixture()
def bucket_on():
    g.set_option("GM_SEED_BUCKET", "1")
    yield
    for k in ("GM_SEED_BUCKET", "GM_HEAVY_MIN", "GM_HEAVY_BUDGET", "GM_VOTE_FIXED"):
        g.set_option(k, None)


def _run(cfg, ix_full, oracle, oix, syn_reads, packed, expect_bucket=True):
    kw = dict(BUCKET_CONFIGS[cfg])
    p = g.Params(**kw)
    B, Q, Ln = packed
    # on a reference this small the k-mer table stops at 12 characters by default; the records need it as long as the seed
    g.set_option("GM_KMER_TABLE", str(p.mer) if p.mer > 12 else None)
    batch = g.Batch(ix_full, len(syn_reads), B.shape[1])
    res = batch.map(p, B, Q, Ln)
    path = batch.path()
    ctr = batch.counters()
    batch.destroy()
    g.set_option("GM_KMER_TABLE", None)
    assert ("k_vote_bucket" in path) == expect_bucket, path
    ores = _oracle_results(oracle, oix, oracle.params(**kw), syn_reads)
    _compare(res, ores, syn_reads)
    return res, ores, ctr


@pytest.mark.parametrize("cfg", list(BUCKET_CONFIGS))
def test_bucket_kernel_matches_oracle(cfg, bucket_on, ix_full, oracle, oix, syn_reads, packed):
    res, ores, ctr = _run(cfg, ix_full, oracle, oix, syn_reads, packed)
    # the work counters: every SA hit the oracle locates, at most the k-mers it searches one by one (configurations without the
    # reference's early exits: --fast, -T, -u stop a read's seed loop on the CPU, the device looks all seeds up)
    if cfg in ("default", "h30", "k3", "m9_j5", "m8_j5", "m8_j5_h31", "m7_j9", "m12_j6", "m14_j7", "m12_j18"):
        assert ctr["sa_hits"] == sum(o["ctr"]["locates"] for o in ores)
        assert ctr["seeds_used"] <= ctr["kmers_searched"] <= sum(o["ctr"]["kmers"] for o in ores)
    if cfg not in ("fast", "fast_m8", "T2", "unique", "m8_j5_h10", "m12_j18"):
        assert len(res["matches"]) > 300


@pytest.mark.parametrize("cfg", ["default", "m8_j5", "m7_j5", "m8_j5_h31", "m7_j9", "m12_j18"])
@pytest.mark.parametrize("opts", [dict(GM_HEAVY_MIN="64"), dict(GM_HEAVY_MIN="8", GM_HEAVY_BUDGET="200000"), dict(GM_VOTE_FIXED="0")],
                         ids=lambda o: ",".join(f"{k[3:]}={v}" for k, v in o.items()))
def test_bucket_kernel_hand_overs(cfg, opts, bucket_on, ix_full, oracle, oix, syn_reads, packed):
    """read x strands handed to the sorted-key path by their own hit counts; candidates through the bump counters instead of own slots"""
    for k, v in opts.items():
        g.set_option(k, v)
    _run(cfg, ix_full, oracle, oix, syn_reads, packed)


def test_bucket_kernel_is_not_chosen_where_it_cannot_run(bucket_on, ix_full, oracle, oix, syn_reads, packed):
    """-k 1 (every hit is a candidate: no filter), more than 32 seeds per strand: the other kernels, same results"""
    BUCKET_CONFIGS["k1_m14"] = dict(min_seed_hits=1, mer=14)
    BUCKET_CONFIGS["m10_j2"] = dict(mer=10, jump=2)
    try:
        _run("k1_m14", ix_full, oracle, oix, syn_reads, packed, expect_bucket=False)
        _run("m10_j2", ix_full, oracle, oix, syn_reads, packed, expect_bucket=False)
    finally:
        del BUCKET_CONFIGS["k1_m14"], BUCKET_CONFIGS["m10_j2"]


def test_switches_are_read_on_every_call(ix_full, syn_reads, packed):
    """gm_set_option: one process, one index, two kernel choices back to back with identical raw results"""
    p = g.Params(mer=8, jump=5)
    B, Q, Ln = packed
    out = []
    for v in ("1", "0"):
        g.set_option("GM_SEED_BUCKET", v)
        batch = g.Batch(ix_full, len(syn_reads), B.shape[1])
        batch.upload(p, B, Q, Ln)
        batch.map_device(p)
        hits, status, self_score, top = batch.raw_hits()
        out.append((batch.path(), hits.tobytes(), status.tobytes(), top.tobytes()))
        batch.destroy()
    g.set_option("GM_SEED_BUCKET", None)
    assert "k_vote_bucket" in out[0][0] and "k_vote_bucket" not in out[1][0]
    assert out[0][1:] == out[1][1:]


@pytest.mark.parametrize("L,T,kw", [(250, None, dict(mer=12, jump=8)),                        # 30 seeds per strand: <8>; 12-mers with an error do not occur: both strands walk
                                    (330, None, dict(mer=12, jump=10, max_kmer_hits=3)),      # > 300 positions per strand: several rounds of 96 questions
                                    (420, 14, dict(mer=14, jump=13))])
def test_bucket_kernel_on_long_reads(L, T, kw, bucket_on, ix_full, oracle, oix, syn_fa):
    """the walk of a strand asks about at most 96 positions per half and round: reads of 250 .. 420 bases need several"""
    kw = dict(kw)
    reads = _long_reads(syn_fa, L, 24, L + 7)
    B, Q, Ln = g.pack_reads([r[1] for r in reads], [r[2] for r in reads])
    p = g.Params(**kw)
    g.set_option("GM_KMER_TABLE", str(T) if T else None)
    try:
        batch = g.Batch(ix_full, len(reads), B.shape[1])
        res = batch.map(p, B, Q, Ln)
        path = batch.path()
        batch.destroy()
    finally:
        g.set_option("GM_KMER_TABLE", None)
    assert "k_vote_bucket" in path, path
    _compare(res, _oracle_results(oracle, oix, oracle.params(**kw), reads), reads)
