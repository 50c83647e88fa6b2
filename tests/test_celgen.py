"""BASELINE configs[0] on the reference's OWN example data (tests/golden/celgen/, made by make_celgen_fixture.py): real C. elegans
sequence under every read (rebuilt from ART's truth alignment examples/Cel_gen.reads.aln), ART's Illumina quality profile (Phred down
to ~6, where the (1 - p) / 3 columns of the PWM matter), 24 869 50-bp reads.  Expected values = what the unmodified reference PROGRAM
wrote for README.md:22's command on the full read file, in default, --no_nw and -b mode.

CPU: the oracle (oracle/gm_oracle.c gmo_run) reproduces the three SAM files and both tracks byte for byte; the mapped positions agree
with ART's truth; the library's index builder writes the same five files as the reference's bwa_index on this real sequence."""
import gzip
import os

import numpy as np
import pytest

from conftest import CELGEN, ROOT

MODES = {"default": (dict(), "sgr"), "no_nw": (dict(nw=0), None), "bs": (dict(mode=1), "gmp")}


def ref_bytes(name):
    return gzip.open(os.path.join(CELGEN, name + ".gz"), "rb").read()


@pytest.fixture(scope="module")
def oix(oracle, celgen):
    return oracle.index_load(celgen[0])


@pytest.mark.parametrize("mode", sorted(MODES))
def test_oracle_equals_reference_program_on_real_sequence(mode, oracle, oix, celgen, tmp_path):
    kw, track = MODES[mode]
    out = str(tmp_path / "o")
    oracle.run(oix, oracle.params(**kw), celgen[1], out, threads=1)
    sam = b"".join(l for l in open(out + ".sam", "rb") if not l.startswith(b"@PG"))
    assert sam == ref_bytes(mode + ".sam"), mode
    if track:
        assert open(out + "." + track, "rb").read() == ref_bytes(f"{mode}.{track}"), (mode, track)


def test_reference_output_agrees_with_art_truth():
    """the fixture is about mapping real reads, not noise: >= 90 % of the reads have a SAM record at ART's position and strand"""
    t = np.load(os.path.join(CELGEN, "truth.npz"))
    names = [l[1:].strip() for i, l in enumerate(gzip.open(os.path.join(CELGEN, "reads.fq.gz"), "rb")) if i % 4 == 0]
    truth = {n: (int(p), int(s)) for n, p, s in zip(names, t["pos"], t["strand"])}
    assert len(truth) == 24869
    hit = set(); mapped = set()
    for l in ref_bytes("default.sam").splitlines():
        if l.startswith(b"@"):
            continue
        f = l.split(b"\t")
        pos, strand = truth[f[0]]
        mapped.add(f[0])
        if abs(int(f[3]) - 1 - pos) <= 3 and (int(f[1]) & 16 != 0) == (strand == 1):
            hit.add(f[0])
    assert len(hit) >= 0.90 * len(truth), (len(hit), len(mapped), len(truth))
    assert len(mapped) < len(truth)                     # and some reads (low quality tails, repeats) are NOT mapped: both branches pinned


@pytest.mark.skipif(not os.path.exists("/root/reference/src"), reason="reference sources only exist in the build container")
def test_index_files_equal_bwa_index_on_real_sequence(celgen, tmp_path):
    """gm_index_build (own SA-IS, gm_index.cpp) against the reference's bwa_index (src/bwtindex.c:187-293) on a real chromosome piece
    (homopolymer runs, tandem repeats): the five files byte for byte"""
    import shutil
    from reflib import RefLib
    fa = str(tmp_path / "celgen.fa")
    shutil.copy(celgen[0], fa)
    RefLib().index_build(fa)
    for ext in ("pac", "ann", "amb", "bwt", "sa"):
        assert open(fa + ".gnumap." + ext, "rb").read() == open(celgen[0] + ".gnumap." + ext, "rb").read(), ext
