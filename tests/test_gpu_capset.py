"""Two shortcuts of the seed walk for seeds longer than the k-mer table (gm_capset.hip), both exact:
  - the set of the k-mers beyond -h: k_seed drops a member with one probe instead of a table lookup + mer - T search steps;
  - the bitmap of the W-mers that occur at all: a k-mer whose last W characters do not occur is dropped (with every k-mer that
    contains them) after one bit.
The seeds - and everything after them - are the same, against the oracle and against the run without the shortcut, and the rank
queries saved show in the work counters.  Both are opt-in (GM_CAPSET=1, GM_KBIT=<W>): at human scale they cost k_seed more than they
save."""
import numpy as np
import pytest

import gnumap_amd as g
from test_gpu_parity import _compare, _oracle_results

pytestmark = pytest.mark.gpu

# (mer, table characters, -h): on the 400 kbp fixture k-mers of 8 .. 10 characters occur 6 .. 0.4 times
CAP_CONFIGS = {
    "m10_T8_h2": dict(T=8, kw=dict(mer=10, jump=5, max_kmer_hits=2)),
    "m10_T6_h1": dict(T=6, kw=dict(mer=10, jump=5, max_kmer_hits=1)),
    "m9_T6_h4": dict(T=6, kw=dict(mer=9, jump=4, max_kmer_hits=4)),
    "m12_T4_h1": dict(T=4, kw=dict(mer=12, jump=6, max_kmer_hits=1)),                 # 8 characters in front of the table's
    "m8_T6_h10_no_nw": dict(T=6, kw=dict(mer=8, jump=5, max_kmer_hits=10, nw=0)),
    "m9_T7_h3_k3": dict(T=7, kw=dict(mer=9, jump=3, max_kmer_hits=3, min_seed_hits=3)),
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


def _map(ix, p, packed, n, opts):
    B, Q, Ln = packed
    for k, v in opts.items():
        g.set_option(k, v)
    batch = g.Batch(ix, n, B.shape[1])
    res = batch.map(p, B, Q, Ln)
    path, ctr = batch.path(), batch.counters()
    batch.destroy()
    for k in opts:
        g.set_option(k, None)
    return res, path, ctr


def _same(a, b):
    for f in ("status", "top_score", "denominator", "match_begin"):
        assert np.array_equal(a[f], b[f]), f
    for arr in ("matches", "positions"):
        for f in a[arr].dtype.names:                            # (the records' padding bytes are not part of the result)
            assert np.array_equal(a[arr][f], b[arr][f]), (arr, f)


@pytest.mark.parametrize("cfg", list(CAP_CONFIGS))
def test_capped_kmer_set_changes_nothing_but_the_work(cfg, ix_full, oracle, oix, syn_reads, packed):
    c = CAP_CONFIGS[cfg]
    p = g.Params(**c["kw"])
    g.set_option("GM_KMER_TABLE", str(c["T"])); g.set_option("GM_SEED_BUCKET", "0"); g.set_option("GM_SEED_FUSED", "0")
    try:
        with_set, path, ctr1 = _map(ix_full, p, packed, len(syn_reads), dict(GM_CAPSET="1"))
        without, _, ctr0 = _map(ix_full, p, packed, len(syn_reads), dict(GM_CAPSET="0"))
    finally:
        for k in ("GM_KMER_TABLE", "GM_SEED_BUCKET", "GM_SEED_FUSED"):
            g.set_option(k, None)
    assert "k_seed" in path, path
    ores = _oracle_results(oracle, oix, oracle.params(**c["kw"]), syn_reads)
    _compare(with_set, ores, syn_reads)
    _same(with_set, without)
    # the same k-mers tried, the same seeds and SA hits; fewer rank queries (the capped k-mers are not searched any more)
    for f in ("kmers_searched", "seeds_used", "sa_hits"):
        assert ctr1[f] == ctr0[f], (f, ctr1[f], ctr0[f])
    assert ctr1["occ_calls"] < ctr0["occ_calls"], (ctr1["occ_calls"], ctr0["occ_calls"])


@pytest.mark.parametrize("cfg,W", [("m10_T8_h2", 10), ("m10_T6_h1", 9), ("m12_T4_h1", 10), ("m9_T7_h3_k3", 8), ("m14_T8", 12), ("m20_T10", 13), ("m16_T12_no_nw", 16)])
def test_bitmap_of_the_kmers_that_occur_changes_nothing_but_the_work(cfg, W, ix_full, oracle, oix, syn_reads, packed):
    extra = {"m14_T8": dict(T=8, kw=dict(mer=14, jump=7)), "m20_T10": dict(T=10, kw=dict(mer=20, jump=10, max_kmer_hits=150)),
             "m16_T12_no_nw": dict(T=12, kw=dict(mer=16, jump=8, nw=0))}
    c = CAP_CONFIGS.get(cfg) or extra[cfg]
    p = g.Params(**c["kw"])
    g.set_option("GM_KMER_TABLE", str(c["T"])); g.set_option("GM_SEED_BUCKET", "0"); g.set_option("GM_SEED_FUSED", "0")
    try:
        with_map, path, ctr1 = _map(ix_full, p, packed, len(syn_reads), dict(GM_KBIT=str(W)))
        without, _, ctr0 = _map(ix_full, p, packed, len(syn_reads), {})
    finally:
        for k in ("GM_KMER_TABLE", "GM_SEED_BUCKET", "GM_SEED_FUSED"):
            g.set_option(k, None)
    assert "k_seed" in path, path
    _compare(with_map, _oracle_results(oracle, oix, oracle.params(**c["kw"]), syn_reads), syn_reads)
    _same(with_map, without)
    for f in ("seeds_used", "sa_hits"):
        assert ctr1[f] == ctr0[f], (f, ctr1[f], ctr0[f])
    assert ctr1["occ_calls"] < ctr0["occ_calls"], (ctr1["occ_calls"], ctr0["occ_calls"])
    assert ctr1["kmers_searched"] >= ctr0["kmers_searched"]          # a dead 18-suffix skips fewer k-mers than the depth the search died at
