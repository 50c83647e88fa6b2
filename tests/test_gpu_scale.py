"""The BASELINE configurations at (or near) their own sizes, each compared read by read with the oracle on a random sample, and the
maximum-size case: a reference with more than 2^31 positions.  The index is built on the box (suffix-array stage on the device) and
loaded into HBM; tools/scale_check.py is the same check as a command-line tool.

    configs[1]  100 Mbp, -m 10 -j 5            -> k_vote_slots<40> (the bench-dominant dense form) against the oracle directly
    configs[2]  156 Mbp, -m 10 -j 5 -h 150     -> the 64-slot (BIG) form
    headline    3.1 Gbp, -m 14 -j 7            -> k_vote_bucket (the kernel and the size bench.py's default line is measured on)
    configs[3]  3.1 Gbp, 150-bp reads, -b      -> k_vote_bucket<6> with bisulfite scoring, .gmp arrays (per-nucleotide track) checked
    configs[4]  3.1 Gbp, --no_nw               -> the hit-count-only path
    repeat-rich 3.1 Gbp, -m 14 -h 150          -> capped k-mers (the walk slides), seeds beyond 28 hits, read x strands handed to the list kernel
"""
import os
import shutil
import sys

import pytest

pytestmark = pytest.mark.gpu
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))


@pytest.fixture(scope="module")
def workdir(tmp_path_factory):
    d = tmp_path_factory.mktemp("scale")
    yield str(d)
    shutil.rmtree(d, ignore_errors=True)


def _ok(out, sample):
    assert out["oracle_sample"] == sample and out["oracle_mismatches"] == 0, out
    assert out["exact_reads_checked"] > 500 and out["exact_reads_origin_found"] == out["exact_reads_checked"], out


def _ok_output(out):
    assert out["output_records"] > 0.5 * out["output_reads"] and out["output_record_mismatch"] == 0, out
    assert out["coverage_bins_checked"] > 100 and out["coverage_bin_mismatches"] == 0 and out["track_totals_ok"], out


def test_configs1_100mbp_m10_dense_vote_form(workdir):
    import scale_check
    out = scale_check.main(["--mbp", "100", "--contigs", "6", "--mer", "10", "--reads", "100000", "--sample", "200", "--steps", "1",
                            "--check-output", "64", "--workdir", workdir])
    _ok(out, 200); _ok_output(out)
    assert 1500 < out["sa_hits_per_read"] < 5000            # ~1700 hits per read x strand: 36 slots -> the 40-slot form


def test_configs2_156mbp_m10_h150_big_vote_form(workdir):
    import scale_check
    out = scale_check.main(["--mbp", "156", "--contigs", "1", "--mer", "10", "--max-kmer-hits", "150", "--reads", "100000", "--sample", "120",
                            "--steps", "1", "--check-output", "48", "--workdir", workdir])
    _ok(out, 120); _ok_output(out)


def test_headline_size_and_kernel_beyond_2_31_positions(workdir):
    """bench.py's default workload at its own size: 3.1 Gbp in 24 contigs, -m 14 -j 7, NW, the bucket-table kernel"""
    import scale_check
    out = scale_check.main(["--mbp", "3100", "--contigs", "24", "--mer", "14", "--reads", "200000", "--sample", "64", "--steps", "1",
                            "--check-output", "32", "--keep", "--workdir", workdir])
    assert out["l_pac"] == 3_100_000_000 and out["l_pac"] > 2 ** 31
    assert "k_vote_bucket<4>" in out["path"], out["path"]
    _ok(out, 64); _ok_output(out)
    assert out["oracle_tail_reads"] >= 10                     # sampled reads that lie beyond 2^31
    assert out["max_reported_pos"] > 2 ** 31
    assert out["vote_retries"] == 0


def test_configs4_no_nw_at_scale(workdir):
    import scale_check
    out = scale_check.main(["--mbp", "3100", "--contigs", "24", "--mer", "14", "--no-nw", "--reads", "200000", "--sample", "48", "--steps", "1",
                            "--check-output", "32", "--keep", "--workdir", workdir])
    _ok(out, 48); _ok_output(out)


def test_configs3_bisulfite_150bp_at_human_size(workdir):
    import scale_check
    out = scale_check.main(["--mbp", "3100", "--contigs", "24", "--mer", "14", "--mode", "1", "--read-len", "150", "--reads", "200000", "--sample", "48",
                            "--steps", "1", "--check-output", "32", "--keep", "--workdir", workdir])
    assert "k_vote_bucket<6>" in out["path"], out["path"]
    _ok(out, 48); _ok_output(out)
    assert out["nuc_bins_checked"] > 1000 and out["nuc_bin_mismatches"] == 0, out


def test_two_kernel_form_at_human_size(workdir):
    """the same index, the k_seed + k_vote_tiny form (what runs where the bucket table does not fit)"""
    import scale_check
    import gnumap_amd as g
    g.set_option("GM_SEED_BUCKET", "0"); g.set_option("GM_SEED_FUSED", "0")
    try:
        out = scale_check.main(["--mbp", "3100", "--contigs", "24", "--mer", "14", "--reads", "100000", "--sample", "48", "--steps", "1", "--keep", "--workdir", workdir])
    finally:
        g.set_option("GM_SEED_BUCKET", None); g.set_option("GM_SEED_FUSED", None)
    assert "k_seed" in out["path"] and "k_vote_tiny" in out["path"], out["path"]
    _ok(out, 48)


def test_repeat_rich_human_size_with_kmer_cap(workdir):
    """28 % of the reference in repeat families, -h 150: capped k-mers make the walk slide, seeds hold up to 150 hits, read x strands
    with too many hits go to the list kernel - all compared with the oracle on a sample"""
    import scale_check
    out = scale_check.main(["--mbp", "3100", "--contigs", "24", "--mer", "14", "--max-kmer-hits", "150", "--repeats", "--reads", "200000", "--sample", "64",
                            "--steps", "1", "--check-output", "32", "--workdir", workdir])
    assert "k_vote_bucket<4>" in out["path"], out["path"]
    assert out["oracle_sample"] == 64 and out["oracle_mismatches"] == 0, out
    assert out["exact_reads_checked"] > 500 and out["exact_reads_origin_found"] >= 0.8 * out["exact_reads_checked"], out      # reads inside repeats end as "too many"
    _ok_output(out)


