"""The oracle (oracle/gm_oracle.c, CPU restatement) against the committed outputs of the REFERENCE's own
functions (tests/golden/ref_vectors.npz, made by tests/golden/make_fixtures.py through oracle/_ref) and
against the known answers of the reference's bin_seq::Test (src/bin_seq.cpp:1046-1127).
Bit-exact everywhere: integers, strings and fp32 score bits."""
import ctypes as C

import numpy as np
import pytest

import os

from conftest import GOLDEN
from reflib import revcomp_pwm, revcomp_str


@pytest.fixture(scope="module")
def ix(oracle, syn_fa):
    return oracle.index_load(syn_fa)


@pytest.fixture(scope="module")
def params(oracle):
    return oracle.params()


def test_index_header(oracle, ix, golden):
    assert ix.contents.seq_len == int(golden["seq_len"])
    assert ix.contents.primary == int(golden["primary"])
    assert ix.contents.n_seqs == 3
    assert [ix.contents.contigs[i].name for i in range(3)] == [b"chrA", b"chrB", b"chrC"]
    assert [ix.contents.contigs[i].offset for i in range(3)] == [0, 150000, 250000]


def test_occ(oracle, ix, golden):
    got = np.array([oracle.lib.gmo_occ(ix, int(k), int(c), None) for k, c in zip(golden["occ_k"], golden["occ_c"])], np.uint64)
    np.testing.assert_array_equal(got, golden["occ_out"])


def test_sa_interval(oracle, ix, golden):
    n_hit = 0
    for kmer, (s, e) in zip(golden["kmers"], golden["kmer_iv"]):
        assert oracle.sa_interval(ix, bytes(kmer)) == (int(s), int(e)), kmer
        n_hit += (s, e) != (0, 0)
    assert n_hit > 300      # the fixture exercises both outcomes


def test_locate(oracle, ix, golden):
    got = np.array([oracle.lib.gmo_locate(ix, int(r), None) for r in golden["loc_rank"]], np.uint64)
    np.testing.assert_array_equal(got, golden["loc_out"])


def test_window(oracle, ix, golden):
    n_empty = 0
    for b, L, w in zip(golden["win_begin"], golden["win_len"], golden["win_out"]):
        assert oracle.window(ix, int(b), int(L)) == bytes(w), (b, L)
        n_empty += len(bytes(w)) == 0
    assert n_empty >= 3     # contig boundary / genome end cases are present


def test_score_table(oracle, params, golden):
    S = np.ctypeslib.as_array(params.S).reshape(256, 4)
    np.testing.assert_array_equal(S.view(np.uint32), golden["S"].view(np.uint32))
    assert params.gap == float(golden["gap"]) and params.max_gap == int(golden["max_gap"])


def test_pwm_and_self_score(oracle, params, golden, syn_reads):
    for i, (name, seq, qual) in enumerate(syn_reads):
        L = len(seq)
        assert L == golden["fq_len"][i]
        P = oracle.pwm(seq, qual)
        np.testing.assert_array_equal(P.view(np.uint32), golden["fq_pwm"][i, :L].view(np.uint32))
        if L > 0:
            s = oracle.lib.gmo_self_score(C.byref(params), np.ascontiguousarray(P), seq, L)
            assert np.float32(s).view(np.uint32) == golden["self_score"][i].view(np.uint32), name


def test_nw_score_and_traceback(oracle, params, golden, syn_reads):
    n_gapped = 0
    for ci in range(len(golden["nw_read"])):
        i = int(golden["nw_read"][ci]); rc = int(golden["nw_rc"][ci]); w = bytes(golden["nw_window"][ci])
        name, seq, qual = syn_reads[i]
        P = oracle.pwm(seq, qual); cons = seq
        if rc:
            P = revcomp_pwm(P); cons = revcomp_str(cons)
        s = oracle.lib.gmo_nw_score(C.byref(params), np.ascontiguousarray(P), len(seq), w)
        assert np.float32(s).view(np.uint32) == golden["nw_score"][ci].view(np.uint32), (name, ci)
        al, n, cg = oracle.traceback(params, P, cons, w)
        assert n == golden["tb_len"][ci]
        assert al.hex().encode() == bytes(golden["tb_aligned_hex"][ci])
        assert cg == bytes(golden["tb_cigar"][ci])
        n_gapped += (b"I" in cg) or (b"D" in cg)
    assert n_gapped > 50


# ---- known answers held by the reference's own unit test, bin_seq::Test (src/bin_seq.cpp:1046-1127) ----
KAT_CONS = b"acgtcgatcgtggctaatcgttcgtagatcgatta"
KAT_GEN1 = b"acgtcgatcgtggctaatcgttgtagatccgatta"
KAT_GEN2 = b"acgtcgtttatcgtggctaatcgttccgattaccc"


def _kat_pwm():
    P = np.zeros((35, 4), np.float32)
    for i, ch in enumerate(KAT_CONS):
        P[i, b"acgt".index(ch)] = 1.0
    return P


def test_kat_traceback(oracle, params):
    al, n, cg = oracle.traceback(params, _kat_pwm(), KAT_CONS, KAT_GEN1)      # bin_seq.cpp:1095-1101
    assert al == b"acgtcgatcgtggctaatcgttggtagat-cgatta" and cg == b"22M1I6M1D6M"
    al, n, cg = oracle.traceback(params, _kat_pwm(), KAT_CONS, KAT_GEN2)      # bin_seq.cpp:1113-1117
    assert al == b"acgtcg---atcgtggctaatcgttcttggatcaatta" and cg == b"6M3D17M1I1M1I4M1I4M"


def test_kat_begin_mid_end_score(oracle, params):
    # bin_seq.cpp:1123-1127: get_align_score(r1, gen_str, 10, 20) == 22*gMATCH + 2*gGAP + 12*gMATCH
    s = oracle.lib.gmo_align_score_be(C.byref(params), _kat_pwm(), 35, KAT_GEN1, 10, 20)
    assert s == np.float32(22 * params.match + 2 * params.gap + 12 * params.match)


def test_cigar_helpers(oracle):
    for cig, fixed, rev in ((b"22M1I6M1D6M", b"22M1I6M1D6M", b"6M1D6M1I22M"), (b"97M3D", b"97M", b"3D97M"), (b"100M", b"100M", b"100M")):
        buf = C.create_string_buffer(cig, 1024)
        oracle.lib.gmo_fix_cigar(buf)
        assert buf.value == fixed
        out = C.create_string_buffer(1024)
        oracle.lib.gmo_reverse_cigar(cig, out)
        assert out.value == rev


def test_output_helpers_against_reference_vectors(oracle, golden):
    """reverse_comp / reverse_CIGAR / fix_CIGAR_for_deletions (inc/SequenceOperations.h:32-123) on every CIGAR the reference's
    traceback produced for the fixture plus edge cases, and on reads / gapped strings (vectors from oracle/_ref)"""
    for cg, fx, rv in zip(golden["hc_in"], golden["hc_fix"], golden["hc_rev"]):
        buf = C.create_string_buffer(bytes(cg), 1024)
        if len(cg):
            oracle.lib.gmo_fix_cigar(buf)
        assert buf.value == bytes(fx), cg
        out = C.create_string_buffer(1024)
        oracle.lib.gmo_reverse_cigar(bytes(cg), out)
        assert out.value == bytes(rv), cg
    oracle.lib.gmo_revcomp_str.argtypes = [C.c_char_p, C.c_int, C.c_char_p]
    for hin, hout in zip(golden["rc_in_hex"], golden["rc_out_hex"]):
        s = bytes.fromhex(hin.decode()); want = bytes.fromhex(hout.decode())
        out = C.create_string_buffer(len(s) + 8)
        oracle.lib.gmo_revcomp_str(s, len(s), out)
        assert out.raw[:len(s)] == want, s


def test_pair_hmm_equals_reference_function(oracle):
    """--snp: bin_seq::pairHMM (src/bin_seq.cpp:60-244; vectors made by tests/golden/make_snp_fixtures.py through the unmodified function):
    the oracle's restatement gives the same 5 floats per window position, bit for bit - fp64 forward / backward matrices, float
    transition products, float accumulation in read order"""
    v = np.load(os.path.join(GOLDEN, "ref_vectors_snp.npz"))
    assert len(v["len"]) >= 80 and (v["strand"] == 1).sum() > 20
    for i in range(len(v["len"])):
        L = int(v["len"][i])
        got = oracle.pair_hmm(v["pwm"][i, :L], bytes(v["cons"][i, :L]), bytes(v["window"][i, :L]))
        np.testing.assert_array_equal(got.view(np.uint32), v["hmm"][i, :L].view(np.uint32), err_msg=f"vector {i}")
    # the posteriors of a window position sum to ~1 at a true locus (the read explains the position), to less off it
    assert 0.99 < float(v["hmm"][0, : int(v["len"][0])].sum(1).mean()) < 1.01
