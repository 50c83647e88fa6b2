"""gm_index_build (gnumap_amd/csrc/gm_index.cpp, from-scratch SA-IS) must write byte-identical
<fa>.gnumap.{pac,ann,amb,bwt,sa} to the reference's own bwa_index (src/bwtindex.c:187): against the committed
reference-built fixture, and — where oracle/_ref is available — against fresh reference builds of random genomes."""
import os
import shutil

import numpy as np
import pytest

import gnumap_amd as g
from reflib import RefLib, have_ref

EXTS = ("pac", "ann", "amb", "bwt", "sa")


def _same(a, b):
    for ext in EXTS:
        with open(f"{a}.gnumap.{ext}", "rb") as fa, open(f"{b}.gnumap.{ext}", "rb") as fb:
            assert fa.read() == fb.read(), ext


def test_build_matches_committed_reference_index(tmp_path, syn_fa):
    fa = str(tmp_path / "syn.fa")
    shutil.copy(syn_fa, fa)
    g.index_build(fa)
    _same(fa, syn_fa)


def _write_random_fasta(path, rng, spec):
    with open(path, "wb") as f:
        for name, n, n_runs, lower in spec:
            seq = np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, n)].copy()
            for _ in range(n_runs):
                p = rng.integers(0, max(1, n - 50)); ln = rng.integers(1, 40)
                seq[p:p + ln] = ord(rng.choice(list("NNNRYn")))
            if lower:
                seq[: n // 3] = np.char.lower(seq[: n // 3].view("S1")).view(np.uint8)
            f.write(b">" + name + b"\n")
            width = int(rng.integers(30, 90))
            for i in range(0, n, width):
                f.write(seq[i:i + width].tobytes() + b"\n")


@pytest.mark.skipif(not have_ref(), reason="oracle/_ref (reference build) not available")
@pytest.mark.parametrize("seed,spec", [
    (1, [(b"one", 1000, 0, False)]),
    (2, [(b"a desc", 4096, 3, True), (b"b", 777, 1, False), (b"c\tx y", 128, 0, False)]),
    (3, [(b"s%d" % i, 50 + 37 * i, i % 3, i % 2 == 0) for i in range(12)]),
    (4, [(b"big", 300_001, 5, True), (b"tail", 3, 0, False)]),
    (5, [(b"x", 128 * 32, 0, False)]),          # seq_len multiple of the occ and SA intervals
    (6, [(b"x", 128 * 32 - 1, 2, False)]),
])
def test_build_matches_fresh_reference_build(tmp_path, seed, spec):
    rng = np.random.default_rng(seed)
    mine = str(tmp_path / "mine.fa"); ref = str(tmp_path / "ref.fa")
    _write_random_fasta(mine, rng, spec)
    shutil.copy(mine, ref)
    g.index_build(mine)
    assert RefLib().index_build(ref) == 0
    _same(mine, ref)


@pytest.mark.skipif(not have_ref(), reason="oracle/_ref (reference build) not available")
@pytest.mark.parametrize("text", [
    b">one desc here\r\nACGTNNACGT\r\nacgtrryACG\r\n>two\r\nTTTTGGGGCCCCAAAA\r\n",          # CRLF
    b">x\nACGTACGTAC\nGGG",                                                                 # no final newline
    b"junk before\n>a  two spaces\n\nACGT\n\n  AC GT\n>b\tq\nNNNNACGTNNNN\n>c\n>d\nA\n",       # blanks inside a line, empty contig
    b">a\nAC>GT\nACGT\n",                                                                   # '>' inside a sequence line
])
def test_fasta_parser_edge_cases_match_reference_build(tmp_path, text):
    mine = str(tmp_path / "mine.fa"); ref = str(tmp_path / "ref.fa")
    open(mine, "wb").write(text); open(ref, "wb").write(text)
    g.index_build(mine)
    assert RefLib().index_build(ref) == 0
    _same(mine, ref)


def test_open_missing_index_fails_loudly(tmp_path):
    fa = str(tmp_path / "none.fa")
    open(fa, "w").write(">x\nACGT\n")
    with pytest.raises(g.GnumapError, match="fail to locate the index files"):
        g.Index(fa, flags=g.GM_INDEX_HOST_ONLY)


def test_build_on_argument_and_device_errors(tmp_path):
    """gm_index_build_on: a bad `where` is an argument error; asking for the device build without a GPU fails loudly (no
    silent host fallback); GM_BUILD_HOST works without one"""
    import torch
    fa = str(tmp_path / "x.fa")
    open(fa, "w").write(">x\n" + "ACGTTGCAAGGCTTAACCGGTTAA" * 20 + "\n")
    with pytest.raises(g.GnumapError):
        g.index_build(fa, where=7)
    if not torch.cuda.is_available():
        with pytest.raises(g.GnumapError, match="no usable HIP device"):
            g.index_build(fa, where=g.GM_BUILD_DEVICE)
    g.index_build(fa, where=g.GM_BUILD_HOST)
    assert os.path.getsize(fa + ".gnumap.bwt") > 0
