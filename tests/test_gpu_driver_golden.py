"""The product (gnumap binary -> C ABI -> HIP kernels) against the outputs of the UNMODIFIED reference program
(tests/golden/ref_runs/, made by tests/golden/make_driver_fixtures.py from oracle/_ref/gnumap_ref): the SAME argv as the
reference was run with, SAM text byte-identical including record order; .sgr / .gmp equal up to the order of fp32 atomic adds."""
import gzip
import json
import os
import subprocess

import pytest

from conftest import GOLDEN, ROOT

pytestmark = pytest.mark.gpu
RUNS = os.path.join(GOLDEN, "ref_runs")
MANIFEST = json.load(open(os.path.join(RUNS, "manifest.json")))
EXE = os.path.join(ROOT, "gnumap_amd", "bin", "gnumap")


def ref_text(mode, ext):
    return gzip.open(os.path.join(RUNS, f"{mode}.{ext}.gz"), "rt").read()


def track(text, ncol):
    d = {}
    for line in text.splitlines():
        f = line.split("\t")
        assert len(f) == ncol, line
        d[(f[0], int(f[1]))] = [float(x) for x in f[2:]]
    return d


def compare_tracks(mine, ref, ncol):
    a, b = track(mine, ncol), track(ref, ncol)
    assert len(b) > 100 or not b
    # a bin whose fp32-atomic sum is a rounding error away from the 0.001 print threshold may differ in presence only
    for k in set(a) ^ set(b):
        assert (a.get(k) or b.get(k))[0] < 2e-3, k
    for k in set(a) & set(b):
        for x, y in zip(a[k], b[k]):
            assert abs(x - y) <= 1e-4 * max(1.0, abs(y)) + 2e-5, (k, x, y)


@pytest.mark.parametrize("extra", [[], ["--locate=sampled"], ["--batch=64", "--workers=2"], ["--chunk_reads=37", "--workers=3"]],
                         ids=["full_sa", "sampled_sa", "batch64", "chunks37"])
@pytest.mark.parametrize("mode", sorted(MANIFEST))
def test_cli_equals_reference_program(mode, extra, tmp_path):
    m = MANIFEST[mode]
    if extra and mode not in ("default", "no_nw", "bs_all", "T2", "u", "illumina", "k1_all", "m16_h150_all", "malformed", "malformed_tail"):
        pytest.skip("index / batching variants run on a subset of the modes")
    out = str(tmp_path / "mine")
    argv = [os.path.join(GOLDEN, a) if a == "subst.txt" else a for a in m["argv"]]
    r = subprocess.run([EXE, "-g", os.path.join(GOLDEN, "syn.fa"), "-o", out, "-a", "0.9"] + argv + extra + [os.path.join(GOLDEN, m["fastq"])],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    sam = "".join(l for l in open(out + ".sam") if not l.startswith("@PG"))
    ref = ref_text(mode, "sam")
    if sam != ref:
        a, b = sam.splitlines(), ref.splitlines()
        first = next((i for i, (x, y) in enumerate(zip(a, b)) if x != y), min(len(a), len(b)))
        pytest.fail(f"{mode}: {len(a)} vs {len(b)} lines, first difference at line {first}:\n  mine {a[first] if first < len(a) else None}\n  ref  {b[first] if first < len(b) else None}")
    if "sgr" in m["tracks"]:
        assert not os.path.exists(out + ".gmp")
        compare_tracks(open(out + ".sgr").read(), ref_text(mode, "sgr"), 3)
    else:
        assert not os.path.exists(out + ".sgr")
        compare_tracks(open(out + ".gmp").read(), ref_text(mode, "gmp"), 8)


def _seed_flags(argv):
    """(mer, jump, -k) a reference command line ends up with (src/Driver.cpp:1163-1171: --fast = -m 14 unless given, jump = mer)"""
    mer, jump, k, fast = 10, None, 2, "--fast" in argv
    for i, x in enumerate(argv):
        if x == "-m": mer = int(argv[i + 1])
        if x == "-j": jump = int(argv[i + 1])
        if x == "-k": k = int(argv[i + 1])
    if fast:
        mer = mer if "-m" in argv else 14
        jump = mer
    return mer, jump if jump else mer // 2, k


@pytest.mark.parametrize("mode", sorted(MANIFEST))
def test_cli_through_the_bucket_kernel_equals_reference_program(mode, tmp_path):
    """k_vote_bucket - the kernel the human-scale headline runs - against the reference PROGRAM's own output, directly: on the 400 kbp
    fixture the default dispatch never picks the k-mer -> positions records (a 10-mer occurs 0.4 times), GM_SEED_BUCKET=1 builds and
    uses them whenever the kernel's preconditions hold (-k >= 2, the table as long as the seed, <= 32 seeds per strand).  The trace line
    of the library proves which kernel ran."""
    m = MANIFEST[mode]
    mer, jump, k = _seed_flags(m["argv"])
    longest = 150 if m["fastq"] == "syn.fq" else 104
    if k < 2 or mer > 15 or (longest - mer + jump - 1) // jump > 32:
        pytest.skip("outside k_vote_bucket's preconditions (the default dispatch covers the mode)")
    out = str(tmp_path / "mine")
    argv = [os.path.join(GOLDEN, a) if a == "subst.txt" else a for a in m["argv"]]
    env = dict(os.environ, GM_SEED_BUCKET="1", GM_TRACE="1")
    if mer > 12:
        env["GM_KMER_TABLE"] = str(mer)              # the table of whole seeds (small references stop at 12 characters by default)
    r = subprocess.run([EXE, "-g", os.path.join(GOLDEN, "syn.fa"), "-o", out, "-a", "0.9"] + argv + [os.path.join(GOLDEN, m["fastq"])],
                       capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    assert "k_vote_bucket<" in r.stderr, r.stderr[-1500:]          # (alone, or behind k_vote_pair for the reads that one flags)
    sam = "".join(l for l in open(out + ".sam") if not l.startswith("@PG"))
    ref = ref_text(mode, "sam")
    if sam != ref:
        a, b = sam.splitlines(), ref.splitlines()
        first = next((i for i, (x, y) in enumerate(zip(a, b)) if x != y), min(len(a), len(b)))
        pytest.fail(f"{mode}: {len(a)} vs {len(b)} lines, first difference at line {first}:\n  mine {a[first] if first < len(a) else None}\n  ref  {b[first] if first < len(b) else None}")
    ext = "sgr" if "sgr" in m["tracks"] else "gmp"
    compare_tracks(open(out + "." + ext).read(), ref_text(mode, ext), 3 if ext == "sgr" else 8)


@pytest.mark.parametrize("mode", ["default", "bs_all", "k1_all", "no_nw"])
def test_cli_with_the_group_traceback_form(mode, tmp_path):
    """reads up to 511 bases take the lane-per-sequence traceback kernel; GM_TRACEBACK=group forces the 8-lane form (the one long
    reads use) on the same inputs: CIGARs, aligned lengths (coverage) and gapped read strings (.gmp) must not change"""
    m = MANIFEST[mode]
    out = str(tmp_path / "mine")
    r = subprocess.run([EXE, "-g", os.path.join(GOLDEN, "syn.fa"), "-o", out, "-a", "0.9"] + m["argv"] + [os.path.join(GOLDEN, m["fastq"])],
                       capture_output=True, text=True, timeout=600, env=dict(os.environ, GM_TRACEBACK="group"))
    assert r.returncode == 0, r.stderr[-2000:]
    assert "".join(l for l in open(out + ".sam") if not l.startswith("@PG")) == ref_text(mode, "sam")
    ext = "sgr" if "sgr" in m["tracks"] else "gmp"
    compare_tracks(open(out + "." + ext).read(), ref_text(mode, ext), 3 if ext == "sgr" else 8)


def _fastq_head(n_reads):
    lines = open(os.path.join(GOLDEN, "syn.fq"), "rb").read().split(b"\n")
    return lines[:4 * n_reads]


@pytest.mark.parametrize("extra", [["--chunk_reads=29"], ["--chunk_reads=29", "--workers=5"], ["--batch=40"], []], ids=["chunks", "chunks_w5", "blocks", "one_chunk"])
@pytest.mark.parametrize("tail", [b"\n", b""], ids=["newline", "no_newline"])
def test_cli_recovers_from_malformed_records_like_the_reference_parser(extra, tail, tmp_path, oracle, syn_fa):
    """SeqReader::get_more_fastq (src/SeqReader.cpp:1091-1144) shifts lines until it is back in step and goes on; the driver does the same
    (chunk mode: the rest of the input is read again in file order from the malformed record on), and nothing of a block behind the
    malformed record reaches the SAM file or the coverage track before that.  Expected output: the oracle's gmo_run, whose reader is
    pinned by the reference program's own output on tests/golden/syn_bad.fq / syn_bad2.fq (test_driver_golden.py)."""
    lines = _fastq_head(200)
    bad = (lines[:4 * 60] + [lines[240], lines[241], b"this is not a plus line", lines[243]] + lines[4 * 61:4 * 120]
           + [lines[480], lines[481], lines[482], lines[483][:30]] + lines[4 * 121:4 * 150] + [b"stray line"] + lines[4 * 150:])
    fq = tmp_path / "bad.fq"
    fq.write_bytes(b"\n".join(bad) + tail)
    out = str(tmp_path / "o"); want = str(tmp_path / "want")
    r = subprocess.run([EXE, "-g", syn_fa, "-o", out, "-a", "0.9"] + extra + [str(fq)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-1500:]
    assert "Trying to recover" in r.stderr
    oracle.run(oracle.index_load(syn_fa), oracle.params(), str(fq), want, threads=1)
    mine = [l for l in open(out + ".sam") if not l.startswith("@PG")]
    ref = [l for l in open(want + ".sam") if not l.startswith("@PG")]
    assert mine == ref and len(mine) > 150
    compare_tracks(open(out + ".sgr").read(), open(want + ".sgr").read(), 3)


def test_cli_refuses_a_read_longer_than_the_kernels_take(tmp_path, syn_fa):
    fq = tmp_path / "long.fq"
    fq.write_bytes(b"\n".join(_fastq_head(3)) + b"\n@long\n" + b"ACGT" * 600 + b"\n+\n" + b"I" * 2400 + b"\n")
    r = subprocess.run([EXE, "-g", syn_fa, "-o", str(tmp_path / "o"), "-a", "0.9", str(fq)], capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "longer than 2048" in r.stderr


def test_cli_empty_and_tiny_inputs(tmp_path):
    out = str(tmp_path / "o")
    empty = tmp_path / "empty.fq"; empty.write_bytes(b"")
    r = subprocess.run([EXE, "-g", os.path.join(GOLDEN, "syn.fa"), "-o", out, "-a", "0.9", str(empty)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-1500:]
    assert [l for l in open(out + ".sam") if not l.startswith("@")] == []
    one = tmp_path / "one.fq"; one.write_bytes(b"\n".join(_fastq_head(1)))          # no trailing newline
    r = subprocess.run([EXE, "-g", os.path.join(GOLDEN, "syn.fa"), "-o", out, "-a", "0.9", "--gpus=1", str(one)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-1500:]
    got = [l for l in open(out + ".sam") if not l.startswith("@")]
    assert got == [l + "\n" for l in ref_text("default", "sam").splitlines() if not l.startswith("@")][:len(got)] and len(got) >= 1


@pytest.mark.parametrize("mode", ["default", "bs_all", "illumina"])
def test_cli_maps_a_too_large_block_in_halves(mode, tmp_path):
    """GM_E_BATCH_TOO_LARGE (more than 2^31 candidates in one block on a repeat-rich reference) makes the driver map the block as two
    halves, recursively; GM_TEST_MAX_BLOCK makes the library report it for every block above 60 reads: same SAM, same tracks"""
    m = MANIFEST[mode]
    out = str(tmp_path / "mine")
    r = subprocess.run([EXE, "-g", os.path.join(GOLDEN, "syn.fa"), "-o", out, "-a", "0.9"] + m["argv"] + [os.path.join(GOLDEN, m["fastq"])],
                       capture_output=True, text=True, timeout=600, env=dict(os.environ, GM_TEST_MAX_BLOCK="60"))
    assert r.returncode == 0, r.stderr[-2000:]
    assert "mapped in halves" in r.stderr
    assert "".join(l for l in open(out + ".sam") if not l.startswith("@PG")) == ref_text(mode, "sam")
    ext = "sgr" if "sgr" in m["tracks"] else "gmp"
    compare_tracks(open(out + "." + ext).read(), ref_text(mode, ext), 3 if ext == "sgr" else 8)
