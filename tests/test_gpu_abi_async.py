"""The enqueue / wait forms of the two batch calls (gm_map_batch_enqueue, gm_output_batch_enqueue, gm_batch_wait) against the
synchronous forms, and what gm_output_batch does with a gm_hits the caller has damaged: refused BEFORE any kernel is enqueued."""
import ctypes as C

import numpy as np
import pytest

import gnumap_amd as g
from gnumap_amd import api

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ix_full(syn_fa):
    return g.Index(syn_fa, flags=g.GM_INDEX_FULL_SA)


@pytest.fixture(scope="module")
def packed(syn_reads):
    return g.pack_reads([r[1] for r in syn_reads], [r[2] for r in syn_reads])


def _fields(a):
    return tuple(np.ascontiguousarray(a[f]).tobytes() for f in a.dtype.names)          # the records' padding bytes are not part of the result


def _snapshot(runner, n, n_recs):
    nm = int(runner.mbegin[n])
    return (runner.status[:n].tobytes(), runner.den[:n].tobytes(), runner.top[:n].tobytes(), runner.mbegin[:n + 1].tobytes(), _fields(runner.matches[:nm]),
            _fields(runner.recs[:n_recs]))


@pytest.mark.parametrize("kw", [{}, dict(nw=0), dict(mode=1), dict(mer=8, jump=5)], ids=["default", "no_nw", "bs", "m8_j5"])
def test_enqueued_calls_equal_the_synchronous_ones(kw, ix_full, packed):
    """three batches in flight from ONE caller thread, blocks of different sizes, each result identical to the synchronous run"""
    B, Q, Ln = packed
    p = g.Params(**kw)
    ix_full.coverage_reset(p.bin_size)
    if p.mode:
        ix_full.coverage_enable_nuc()
    n = len(Ln)
    cuts = [(0, n // 3), (n // 3, 2 * n // 3 + 7), (2 * n // 3 + 7, n)]
    Bp = api.pinned_empty(B.shape, np.uint8); Qp = api.pinned_empty(Q.shape, np.uint8); Lp = api.pinned_empty(n, np.uint16)
    Bp[:] = B; Qp[:] = Q; Lp[:] = Ln
    want = []
    sync = api.BlockRunner(ix_full, p, n, B.shape[1])
    for lo, hi in cuts:
        m, nr = sync.run(Bp[lo:hi], Qp[lo:hi], Lp[lo:hi])
        want.append((m, nr, _snapshot(sync, hi - lo, nr)))
    cov_sync = ix_full.coverage_download().copy()
    ix_full.coverage_reset(p.bin_size)
    if p.mode:
        ix_full.coverage_enable_nuc()
    runners = [api.BlockRunner(ix_full, p, n, B.shape[1]) for _ in cuts]
    for r, (lo, hi) in zip(runners, cuts):
        r.run_async(Bp[lo:hi], Qp[lo:hi], Lp[lo:hi])
    for r, (lo, hi), (m, nr, snap) in zip(runners, cuts, want):
        got_m, got_nr = r.wait()
        assert (got_m, got_nr) == (m, nr)
        assert _snapshot(r, hi - lo, nr) == snap
    np.testing.assert_allclose(ix_full.coverage_download(), cov_sync, rtol=1e-5, atol=1e-5)      # three deposits in another order: fp32 atomics
    ix_full.coverage_reset(8)


def test_wait_reports_the_failing_call_and_skips_what_was_queued_behind_it(ix_full, packed):
    B, Q, Ln = packed
    p = g.Params()
    r = api.BlockRunner(ix_full, p, len(Ln), B.shape[1])
    r.matches = api.pinned_empty(4, api.MATCH_DTYPE)                     # far too small: gm_map_batch answers GM_E_CAPACITY, the output call behind it is skipped
    r.run_async(B, Q, Ln)
    assert g.lib().gm_batch_wait(r.batch.h) == api.GM_E_CAPACITY
    assert int(r._h.matches_cap) > 400                                    # the sizes needed were written back
    assert g.lib().gm_batch_wait(r.batch.h) == 0                          # the error has been handed over once
    r._alloc_hits(int(r._h.matches_cap) + 64, int(r._h.positions_cap) + 64)
    m, nr = r.run(B, Q, Ln)
    assert m > 400 and nr > 400


def test_damaged_hits_are_refused_before_any_kernel_runs(ix_full, packed):
    """an edited match with a read index of another read / a position range beyond the buffer used to reach k_out_items and
    k_traceback before the host noticed"""
    B, Q, Ln = packed
    p = g.Params()
    batch = g.Batch(ix_full, len(Ln), B.shape[1])
    res = batch.map(p, B, Q, Ln)
    recs, _ = batch.output(p, res)
    assert len(recs) > 400
    L = g.lib()

    def out_rc():
        so = api.gm_sam_out()
        recs_ = np.zeros(4 * len(Ln), api.SAM_DTYPE); pool = np.zeros(64 * len(Ln), np.uint8)
        so.recs = recs_.ctypes.data; so.recs_cap = len(recs_); so.cigar_pool = pool.ctypes.data; so.cigar_cap = len(pool)
        return L.gm_output_batch(ix_full.h, C.byref(p.c), batch.h, C.byref(res["_reads"]), C.byref(res["_struct"]), C.byref(so), None)

    M = res["matches"]
    keep = M[5].copy()
    # the stamp of gm_map_batch says "use what is resident in HBM": with it, edits of the host copy are not even looked at
    M[5]["read"] = 10 ** 9
    assert out_rc() == 0
    M[5] = keep
    res["_struct"].stamp = 0                                              # a caller that edits the records says so
    for field, value in (("read", 10 ** 9), ("read", int(M[5]["read"]) + 1), ("pos_end", 2 ** 31), ("first_strand", 7), ("first_pos", 2 ** 40)):
        M[5][field] = value
        assert out_rc() == -1 and "gm_hits" in L.gm_last_error().decode(), field
        M[5] = keep
    P = res["positions"]
    keep_p = P[3].copy()
    P[3]["pos"] = 2 ** 40
    assert out_rc() == -1
    P[3] = keep_p
    assert out_rc() == 0                                                  # intact again: accepted (uploaded again, validated)
    recs2, _ = batch.output(p, res)
    assert recs2.tobytes() == recs.tobytes() or [tuple(r[f] for f in recs.dtype.names) for r in recs2] == [tuple(r[f] for f in recs.dtype.names) for r in recs]
    batch.destroy()
