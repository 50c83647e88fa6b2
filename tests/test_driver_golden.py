"""Driver-level pin of the oracle: oracle/gm_oracle.c::gmo_run against the outputs of the UNMODIFIED reference program
(oracle/_ref/gnumap_ref = src/Driver.cpp, inc/align_seq2_raw.cpp, inc/ScoredSeq.h, src/*ScoredSeq.cpp, src/GenomeBwt.cpp ...)
committed under tests/golden/ref_runs/ by tests/golden/make_driver_fixtures.py.

Byte-identical SAM text INCLUDING record order (-c 1 order), byte-identical .sgr / .gmp text, in every mode of the manifest:
adaptive k-mer walk, votes, accept test, unique map, denominator, winner rule, MAPQ, CIGAR, XA/XP/X0 and the coverage tracks."""
import gzip
import json
import os
import subprocess

import pytest

from conftest import GOLDEN, ROOT

RUNS = os.path.join(GOLDEN, "ref_runs")
MANIFEST = json.load(open(os.path.join(RUNS, "manifest.json")))


def ref_text(mode, ext):
    return gzip.open(os.path.join(RUNS, f"{mode}.{ext}.gz"), "rb").read()


@pytest.fixture(scope="module")
def oix(oracle, syn_fa):
    return oracle.index_load(syn_fa)


@pytest.mark.parametrize("mode", sorted(MANIFEST))
def test_oracle_run_equals_reference_program(mode, oracle, oix, tmp_path):
    m = MANIFEST[mode]
    out = str(tmp_path / "o")
    kw = dict(m["params"]); subst = kw.pop("_subst", None)
    p = oracle.params(**kw)
    if subst:
        oracle.apply_subst(p, os.path.join(GOLDEN, subst))
    st = oracle.run(oix, p, os.path.join(GOLDEN, m["fastq"]), out, threads=1)
    sam = b"".join(l for l in open(out + ".sam", "rb") if not l.startswith(b"@PG"))
    assert sam == ref_text(mode, "sam"), mode
    assert sam.count(b"\n") == m["sam_lines"]
    for ext in ("sgr", "gmp"):
        if ext in m["tracks"]:
            assert open(out + "." + ext, "rb").read() == ref_text(mode, ext), (mode, ext)
        else:
            assert not os.path.exists(out + "." + ext)
    assert st.n_records == m["sam_lines"] - 3           # 3 @SQ lines


def test_oracle_threads_give_the_same_records(oracle, oix, tmp_path):
    out = str(tmp_path / "o")
    oracle.run(oix, oracle.params(), os.path.join(GOLDEN, "syn.fq"), out, threads=4)
    sam = b"".join(l for l in open(out + ".sam", "rb") if not l.startswith(b"@PG"))
    assert sam == ref_text("default", "sam")


@pytest.mark.skipif(not os.path.exists(os.path.join(ROOT, "oracle", "_ref", "gnumap_ref")), reason="reference program only exists in the build container")
@pytest.mark.parametrize("mode", ["default", "bs_all", "illumina"])
def test_fixtures_are_what_the_reference_program_writes_now(mode, tmp_path):
    """the committed fixtures are reproducible: re-run the reference program and compare"""
    import shutil
    m = MANIFEST[mode]
    for f in os.listdir(GOLDEN):
        if f.startswith("syn"):
            shutil.copy(os.path.join(GOLDEN, f), tmp_path)
    r = subprocess.run([os.path.join(ROOT, "oracle", "_ref", "gnumap_ref"), "-g", "syn.fa", "-o", "r", "-a", "0.9"] + m["argv"] + [m["fastq"]],
                       cwd=tmp_path, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-800:]
    sam = b"".join(l for l in open(tmp_path / "r.sam", "rb") if not l.startswith(b"@PG"))
    assert sam == ref_text(mode, "sam")
    for ext in m["tracks"]:
        assert open(tmp_path / f"r.{ext}", "rb").read() == ref_text(mode, ext)


def test_oracle_snp_mode_keeps_the_default_sam(oracle, oix, tmp_path):
    """--snp (SNPScoredSeq) changes the deposit, not the mapping: the reference program's SAM with --snp is its default-mode SAM (it
    then aborts in PrintFinalSNP on gsl_cdf_chisq_P - GSL is not in the image - so its .gmp cannot be a fixture; the deposit is pinned at
    function level, tests/test_oracle_golden.py::test_pair_hmm_equals_reference_function).  The oracle's .gmp: one line per covered
    position, the five per-nucleotide sums of a position adding up to about its coverage."""
    out = str(tmp_path / "o")
    oracle.run(oix, oracle.params(mode=5), os.path.join(GOLDEN, "syn.fq"), out, threads=2)
    sam = b"".join(l for l in open(out + ".sam", "rb") if not l.startswith(b"@PG"))
    assert sam == ref_text("default", "sam")
    assert not os.path.exists(out + ".sgr")
    rows = [l.split("\t") for l in open(out + ".gmp")]
    assert len(rows) > 30000 and all(len(r) == 8 for r in rows)
    ratio = [sum(float(x) for x in r[3:]) / float(r[2]) for r in rows[::50]]
    assert 0.9 < sorted(ratio)[len(ratio) // 2] < 1.05
