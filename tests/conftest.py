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
