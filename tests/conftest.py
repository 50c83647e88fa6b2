import os
import subprocess
import sys

import pytest

try:                                    # torch bundles its own HIP runtime: when a test uses both torch.cuda and libgnumap_hip
    import torch  # noqa: F401          # in one process, torch has to be loaded first (bench.py does the same)
except Exception:                       # pragma: no cover
    torch = None

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session", autouse=True)
def _build_checkers():
    """Build the test-only checkers (oracle restatement; reference harness where /root/reference exists)."""
    subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "oracle", "ref", "refbin"], check=True,
                   stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    # the product itself (hipcc cross-compiles gfx950 without a GPU); a failure here must be loud
    subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "gnumap_amd")], check=True, stdout=subprocess.DEVNULL)


@pytest.fixture(scope="session")
def golden():
    import numpy as np
    return np.load(os.path.join(GOLDEN, "ref_vectors.npz"))


@pytest.fixture(scope="session")
def oracle():
    from reflib import OracleLib
    return OracleLib()


@pytest.fixture(scope="session")
def syn_fa():
    return os.path.join(GOLDEN, "syn.fa")


@pytest.fixture(scope="session")
def syn_fq():
    return os.path.join(GOLDEN, "syn.fq")


def read_fastq(path):
    recs = []
    with open(path, "rb") as f:
        while True:
            name = f.readline()
            if not name:
                break
            seq = f.readline().rstrip(b"\n"); f.readline(); qual = f.readline().rstrip(b"\n")
            recs.append((name[1:].rstrip(b"\n").decode(), seq, qual))
    return recs


@pytest.fixture(scope="session")
def syn_reads(syn_fq):
    return read_fastq(syn_fq)


CELGEN = os.path.join(GOLDEN, "celgen")


@pytest.fixture(scope="session")
def celgen(tmp_path_factory):
    """the real-sequence fixture (tests/golden/celgen/, BASELINE configs[0]): genome + reads unpacked into a session directory and indexed
    there with the library's own builder (host SA-IS: no device needed; byte-compatible with bwa_index, tests/test_index_build.py).
    Returns (fasta path, fastq path)."""
    import gzip
    import shutil
    import gnumap_amd as g
    d = tmp_path_factory.mktemp("celgen")
    fa, fq = str(d / "celgen.fa"), str(d / "reads.fq")
    for src, dst in ((os.path.join(CELGEN, "celgen.fa.gz"), fa), (os.path.join(CELGEN, "reads.fq.gz"), fq)):
        with gzip.open(src, "rb") as f, open(dst, "wb") as o:
            shutil.copyfileobj(f, o)
    g.index_build(fa, g.GM_BUILD_HOST)
    return fa, fq
