"""k_nw_rows (gm_nw.hip): the DP kernel for blocks of ONE read length - the read row and the window brought into DP order at load time,
row values from a per-workgroup LDS table, band columns as lane masks.  Its score bits must be k_nw_lane's and the oracle's for every
length (the chunk structure changes with L mod 8 and L / 8), every window alignment (the funnel shift of the 2-bit window), both
strands, every scoring mode and both Phred tables; a block with two lengths or a quality character above 127 must keep k_nw_lane."""
import ctypes as C

import numpy as np
import pytest

import gnumap_amd as g
from reflib import revcomp_pwm
from test_gpu_parity import _compare, _oracle_results

pytestmark = pytest.mark.gpu
LENGTHS = [24, 25, 31, 32, 33, 39, 40, 41, 47, 48, 50, 56, 63, 64, 65, 72, 75, 76, 96, 97, 99, 100, 101, 104, 105, 111, 112, 113, 120, 125, 136, 143, 144, 145, 149, 150]


@pytest.fixture(scope="module")
def ix_full(syn_fa):
    return g.Index(syn_fa, flags=g.GM_INDEX_FULL_SA)


@pytest.fixture(scope="module")
def oix(oracle, syn_fa):
    return oracle.index_load(syn_fa)


def _cut(syn_reads, L):
    """the fixture's reads cut to exactly L bases (the 150-base reads for L > 100)"""
    return [(n, s[:L], q[:L]) for n, s, q in syn_reads if len(s) >= L]


@pytest.mark.parametrize("L", LENGTHS)
def test_score_bits_for_every_length_and_alignment(L, ix_full, oracle, oix, syn_reads):
    """function level (gm_dev_nw_score): every read against windows at 40 consecutive starts (all 16 phases of the packed reference
    word and both neighbours), both strands: fp32 bits equal to the oracle's get_align_score and to k_nw_lane's"""
    reads = _cut(syn_reads, L)[:24]
    B, Q, Ln = g.pack_reads([r[1] for r in reads], [r[2] for r in reads])
    p = g.Params(); op = oracle.params()
    rng = np.random.default_rng(L)
    ridx, strand, pos = [], [], []
    for k in range(len(reads)):
        base = int(rng.integers(0, 140000 - L))
        for d in range(40):
            ridx.append(k); strand.append((k + d) & 1); pos.append(base + d)
    ridx = np.array(ridx, np.uint32); strand = np.array(strand, np.uint8); pos = np.array(pos, np.uint64)
    score, valid = ix_full.dev_nw_score(p, B, Q, Ln, ridx, strand, pos)
    g.set_option("GM_NW", "lane")
    try:
        score_lane, valid_lane = ix_full.dev_nw_score(p, B, Q, Ln, ridx, strand, pos)
    finally:
        g.set_option("GM_NW", None)
    assert valid.all() and valid_lane.all()
    np.testing.assert_array_equal(score.view(np.uint32), score_lane.view(np.uint32))
    for k in range(0, len(ridx), 7):
        _, seq, qual = reads[ridx[k]]
        P = oracle.pwm(seq, qual)
        if strand[k]:
            P = revcomp_pwm(P)
        w = oracle.window(oix, int(pos[k]), L)
        want = np.float32(oracle.lib.gmo_nw_score(C.byref(op), np.ascontiguousarray(P, np.float32), L, w))
        assert score[k].view(np.uint32) == want.view(np.uint32), (L, k)


MODES = {"default": {}, "bs": dict(mode=1), "b2": dict(mode=2), "atog": dict(mode=3), "gap6": dict(gap=-6.0), "k1_m14": dict(min_seed_hits=1, mer=14), "a07": dict(align_score=0.7)}


@pytest.mark.parametrize("L", [50, 97, 100, 150])
@pytest.mark.parametrize("mode", sorted(MODES))
def test_whole_path_with_uniform_blocks(L, mode, ix_full, oracle, oix, syn_reads):
    """gm_map_batch on blocks of one length: the path string names k_nw_rows, the results are the oracle's read by read"""
    reads = _cut(syn_reads, L)
    kw = MODES[mode]
    p = g.Params(**kw); op = oracle.params(**kw)
    B, Q, Ln = g.pack_reads([r[1] for r in reads], [r[2] for r in reads])
    batch = g.Batch(ix_full, len(reads), B.shape[1])
    res = batch.map(p, B, Q, Ln)
    assert "nw=k_nw_rows" in batch.path(), batch.path()
    _compare(res, _oracle_results(oracle, oix, op, reads), reads)
    g.set_option("GM_NW", "lane")
    try:
        res2 = batch.map(p, B, Q, Ln)
        assert "nw=k_nw_lane" in batch.path()
    finally:
        g.set_option("GM_NW", None)
    for key in ("status", "top_score", "denominator"):
        np.testing.assert_array_equal(res[key], res2[key])
    batch.destroy()


def test_illumina_table_and_fallback_point(ix_full, oracle, oix, syn_reads):
    """--illumina: reads before the first quality below '@' use the Phred+64 table, the rest Phred+33 (SeqReader.cpp:1171-1180): both
    value tables are resident and picked per read"""
    reads = _cut(syn_reads, 100)[:200]
    shifted = [(n, s, bytes(min(126, c + 31) for c in q)) for n, s, q in reads[:120]] + reads[120:]       # Phred+64 until read 120
    p = g.Params(illumina=1); op64 = oracle.params(illumina=1); op33 = oracle.params()
    B, Q, Ln = g.pack_reads([r[1] for r in shifted], [r[2] for r in shifted])
    batch = g.Batch(ix_full, len(shifted), B.shape[1])
    res = batch.map(p, B, Q, Ln)
    assert "nw=k_nw_rows" in batch.path(), batch.path()
    ores = [oracle.map_read(oix, op64 if i < 120 else op33, oracle.pwm(s_, q_, illumina=1 if i < 120 else 0), s_) for i, (n_, s_, q_) in enumerate(shifted)]
    _compare(res, ores, shifted)
    batch.destroy()


def test_blocks_the_kernel_must_not_take(ix_full, oracle, oix, syn_reads):
    """two lengths in a block: k_nw_lane, same results as the oracle.  A quality character above 127 is a negative Phred value for the
    reference (signed char) and an error for both (GM_E_BAD_QUAL): the value table, which stops at 127, is never asked about it"""
    reads = _cut(syn_reads, 100)[:120]
    mixed = reads[:60] + [(n, s[:99], q[:99]) for n, s, q in reads[60:]]
    high = [(n, s, q[:10] + bytes([200]) + q[11:]) if i % 9 == 0 else (n, s, q) for i, (n, s, q) in enumerate(reads)]
    p = g.Params(); op = oracle.params()
    B, Q, Ln = g.pack_reads([r[1] for r in mixed], [r[2] for r in mixed])
    batch = g.Batch(ix_full, len(mixed), B.shape[1])
    res = batch.map(p, B, Q, Ln)
    assert "nw=k_nw_lane" in batch.path(), batch.path()
    _compare(res, _oracle_results(oracle, oix, op, mixed), mixed)
    B, Q, Ln = g.pack_reads([r[1] for r in high], [r[2] for r in high])
    with pytest.raises(g.GnumapError):
        batch.map(p, B, Q, Ln)
    batch.destroy()
