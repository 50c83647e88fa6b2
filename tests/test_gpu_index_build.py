"""Device index build (gnumap_amd/csrc/gm_sa_build.hip: prefix doubling over radix sorts in HBM) must write the same bytes as
the host SA-IS build (gm_index.cpp), which tests/test_index_build.py pins to the reference's bwa_index (src/bwtindex.c:187):
random genomes with ambiguity runs, highly repetitive genomes (many doubling rounds), degenerate and tiny inputs, the
committed reference-built fixture, and a 30 Mbp genome."""
import shutil

import numpy as np
import pytest

import gnumap_amd as g

pytestmark = pytest.mark.gpu
EXTS = ("pac", "ann", "amb", "bwt", "sa")
ACGT = np.frombuffer(b"ACGT", np.uint8)


def _same(a, b):
    for ext in EXTS:
        with open(f"{a}.gnumap.{ext}", "rb") as fa, open(f"{b}.gnumap.{ext}", "rb") as fb:
            assert fa.read() == fb.read(), ext


def _write(path, contigs, width=70):
    with open(path, "wb") as f:
        for name, seq in contigs:
            f.write(b">" + name + b"\n")
            for i in range(0, len(seq), width):
                f.write(bytes(seq[i:i + width]) + b"\n")


def _both(tmp_path, contigs):
    host = str(tmp_path / "host.fa"); dev = str(tmp_path / "dev.fa")
    _write(host, contigs)
    shutil.copy(host, dev)
    g.index_build(host, where=g.GM_BUILD_HOST)
    g.index_build(dev, where=g.GM_BUILD_DEVICE)
    _same(host, dev)


def _rand(rng, n):
    return ACGT[rng.integers(0, 4, n)].copy()


def test_random_genome_with_ambiguity_runs(tmp_path):
    rng = np.random.default_rng(1)
    a = _rand(rng, 200_003); a[1000:1040] = ord("N"); a[5:6] = ord("R"); a[150_000:150_500] = ord("n")
    _both(tmp_path, [(b"a desc", a), (b"b", _rand(rng, 777)), (b"c", _rand(rng, 128 * 32))])


def test_repetitive_genome_needs_many_rounds(tmp_path):
    rng = np.random.default_rng(2)
    unit = _rand(rng, 37)
    tandem = np.tile(unit, 3000)                               # 111 kb tandem repeat: LCPs up to ~111 k
    dup = _rand(rng, 60_000)
    seq = np.concatenate([_rand(rng, 5000), tandem, dup, _rand(rng, 100), dup, tandem[:50_000], dup[::-1]])
    _both(tmp_path, [(b"rep", seq), (b"polyA", np.full(20_000, ord("A"), np.uint8)), (b"ac", np.tile(np.frombuffer(b"AC", np.uint8), 9000))])


@pytest.mark.parametrize("seq", [b"A", b"ACGT", b"TTTTTTTTTTTTTTTTTTTTTTTTT", b"GATTACAGATTACAGATTACA", b"ACGTACGTACGTACGTACGTACGTACGTACGTACGTACGTACG"])
def test_tiny_inputs(tmp_path, seq):
    _both(tmp_path, [(b"t", np.frombuffer(seq, np.uint8))])


def test_interval_boundaries(tmp_path):
    rng = np.random.default_rng(3)
    for n in (128 * 32, 128 * 32 - 1, 128 * 32 + 1, 16 * 7, 31, 32, 33):
        d = tmp_path / f"n{n}"
        d.mkdir()
        _both(d, [(b"x", _rand(rng, n))])


def test_committed_reference_built_fixture(tmp_path, syn_fa):
    fa = str(tmp_path / "syn.fa")
    shutil.copy(syn_fa, fa)
    g.index_build(fa, where=g.GM_BUILD_DEVICE)
    _same(fa, syn_fa)


def test_30mbp_genome(tmp_path):
    rng = np.random.default_rng(4)
    _both(tmp_path, [(b"c%d" % i, _rand(rng, n)) for i, n in enumerate((12_000_000, 9_999_999, 8_000_001))])
