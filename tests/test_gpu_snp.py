"""--snp (SNPScoredSeq, src/SNPScoredSeq.cpp:25-109) on the device: the pair HMM of every kept sequence (bin_seq::pairHMM,
src/bin_seq.cpp:60-244) bit for bit against the vectors the UNMODIFIED reference function produced (tests/golden/ref_vectors_snp.npz),
and the deposit - coverage + the five per-nucleotide tracks at every place of a kept sequence, the other strand through
reverse_comp_cpy_phmm - against the oracle's restatement of SNPScoredSeq::score, through the C ABI and through the driver binary.
The likelihood-ratio columns of the reference's .gmp need GSL (absent here): not produced, not compared."""
import gzip
import os
import subprocess

import numpy as np
import pytest

import gnumap_amd as g
from conftest import GOLDEN, ROOT
from test_gpu_driver_golden import compare_tracks

pytestmark = pytest.mark.gpu
GM_MODE_SNP = 5


@pytest.fixture(scope="module")
def ix_full(syn_fa):
    return g.Index(syn_fa, flags=g.GM_INDEX_FULL_SA)


def test_pair_hmm_bits_equal_reference_function(ix_full, syn_reads):
    v = np.load(os.path.join(GOLDEN, "ref_vectors_snp.npz"))
    B, Q, Ln = g.pack_reads([r[1] for r in syn_reads], [r[2] for r in syn_reads])
    out = ix_full.dev_pair_hmm(g.Params(mode=GM_MODE_SNP), B, Q, Ln, v["read"].astype(np.uint32), v["strand"].astype(np.uint8), v["begin"].astype(np.uint64))
    for i in range(len(v["len"])):
        L = int(v["len"][i])
        np.testing.assert_array_equal(out[i, :L].view(np.uint32), v["hmm"][i, :L].view(np.uint32), err_msg=f"vector {i} (read {v['read'][i]}, strand {v['strand'][i]})")


def test_deposit_through_the_abi_matches_oracle(ix_full, oracle, syn_fa, syn_reads):
    reads = syn_reads[:160]
    p = g.Params(mode=GM_MODE_SNP); op = oracle.params(mode=GM_MODE_SNP)
    assert p.bin_size == 1
    oix = oracle.index_load(syn_fa)
    B, Q, Ln = g.pack_reads([r[1] for r in reads], [r[2] for r in reads])
    ix_full.coverage_reset(1); ix_full.coverage_enable_nuc()
    batch = g.Batch(ix_full, len(reads), B.shape[1])
    res = batch.map(p, B, Q, Ln)
    recs, cig = batch.output(p, res)
    cov = ix_full.coverage_download(); nuc = ix_full.coverage_download_nuc().reshape(5, -1)
    # the same records as the default mode (the mapping does not know about --snp)
    ix_full.coverage_reset(1)
    res0 = batch.map(g.Params(), B, Q, Ln); recs0, cig0 = batch.output(g.Params(bin_size=1), res0)
    assert recs.tobytes() == recs0.tobytes() and list(cig) == list(cig0)
    want_cov = np.zeros_like(cov); want_nuc = np.zeros_like(nuc)
    n_dep = 0
    for name, seq, qual in reads:
        st, _, deps = oracle.read_output(oix, op, oracle.pwm(seq, qual), seq)
        for pos, span, w, hmm in deps:
            assert hmm is not None and hmm.shape == (span, 5) and span == len(seq)
            w = np.float32(w)
            want_cov[pos:pos + span] += w
            want_nuc[:, pos:pos + span] += (hmm * w).T
            n_dep += 1
    assert n_dep > 100
    np.testing.assert_allclose(cov, want_cov, rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(nuc, want_nuc, rtol=1e-4, atol=2e-5)
    # the five tracks of a covered position add up to about its coverage (the posteriors of a position sum to ~1 at a true locus)
    covered = want_cov > 0.5
    assert covered.sum() > 5000 and np.median(nuc[:, covered].sum(0) / cov[covered]) > 0.95
    batch.destroy()
    ix_full.coverage_reset(8)


def test_cli_snp_mode(tmp_path, oracle, syn_fa, syn_fq):
    exe = os.path.join(ROOT, "gnumap_amd", "bin", "gnumap")
    out = str(tmp_path / "mine"); want = str(tmp_path / "want")
    r = subprocess.run([exe, "-g", syn_fa, "-o", out, "-a", "0.9", "--snp", syn_fq], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-1500:]
    # SAM: what the reference program writes with --snp before it aborts in PrintFinalSNP = its default-mode SAM (tests/golden/ref_runs)
    sam = "".join(l for l in open(out + ".sam") if not l.startswith("@PG"))
    assert sam == gzip.open(os.path.join(GOLDEN, "ref_runs", "default.sam.gz"), "rt").read()
    oracle.run(oracle.index_load(syn_fa), oracle.params(mode=GM_MODE_SNP), syn_fq, want, threads=1)
    assert not os.path.exists(out + ".sgr")
    compare_tracks(open(out + ".gmp").read(), open(want + ".gmp").read(), 8)
