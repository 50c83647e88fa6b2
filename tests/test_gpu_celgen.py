"""BASELINE configs[0] on the reference's own example data through the PRODUCT (gnumap binary -> C ABI -> HIP kernels): SAM text
byte-identical to what the unmodified reference program wrote (tests/golden/celgen/), tracks equal up to the order of fp32 atomic adds.
At 5 Mbp / -m 10 a k-mer occurs ~4.7 times: the DEFAULT dispatch picks k_vote_bucket here, so the kernel of the human-scale headline is
pinned to the reference program directly, on real sequence with ART's quality profile (Phred down to ~6)."""
import gzip
import os
import subprocess

import pytest

from conftest import CELGEN, ROOT
from test_gpu_driver_golden import compare_tracks

pytestmark = pytest.mark.gpu
EXE = os.path.join(ROOT, "gnumap_amd", "bin", "gnumap")
MODES = {"default": ([], "sgr"), "no_nw": (["--no_nw"], None), "bs": (["-b"], "gmp")}


def ref_text(name):
    return gzip.open(os.path.join(CELGEN, name + ".gz"), "rt").read()


@pytest.mark.parametrize("variant", ["default_dispatch", "sampled_sa", "no_bucket", "chunks"])
@pytest.mark.parametrize("mode", sorted(MODES))
def test_cli_equals_reference_program_on_real_sequence(mode, variant, celgen, tmp_path):
    flags, track = MODES[mode]
    fa, fq = celgen
    extra, env = [], dict(os.environ, GM_TRACE="1")
    if variant == "sampled_sa":
        extra = ["--locate=sampled"]                     # the faithful locate: LF walks to the sampled ranks (src/bwt.c:86-96)
    elif variant == "no_bucket":
        env["GM_SEED_BUCKET"] = "0"                      # the round-2 forms (k-mer table inside k_vote_tiny / k_seed)
    elif variant == "chunks":
        extra = ["--chunk_reads=3000", "--workers=4"]
    out = str(tmp_path / "mine")
    r = subprocess.run([EXE, "-g", fa, "-o", out, "-a", "0.9"] + flags + extra + [fq], capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    if variant in ("default_dispatch", "chunks"):
        assert "k_vote_bucket<" in r.stderr, r.stderr[-1500:]          # (alone, or behind k_vote_pair for the reads that one flags)
    else:
        assert "k_vote_bucket<" not in r.stderr
    sam = "".join(l for l in open(out + ".sam") if not l.startswith("@PG"))
    ref = ref_text(mode + ".sam")
    if sam != ref:
        a, b = sam.splitlines(), ref.splitlines()
        first = next((i for i, (x, y) in enumerate(zip(a, b)) if x != y), min(len(a), len(b)))
        pytest.fail(f"{mode}/{variant}: {len(a)} vs {len(b)} lines, first difference at line {first}:\n  mine {a[first] if first < len(a) else None}\n  ref  {b[first] if first < len(b) else None}")
    if track:
        compare_tracks(open(out + "." + track).read(), ref_text(f"{mode}.{track}"), 3 if track == "sgr" else 8)
