"""Multi-GPU readiness on whatever the box has: the driver binary with --gpus = every visible device (one index replica + workers per
GPU, RCCL all-reduce of the coverage track at the end: src/Driver.cpp:1660-1672), gm_coverage_allreduce on N handles, and bench.py's own
all-reduce leg (real coverage of one block per rank + a known pattern, totals checked) rehearsed through torch.distributed.  On a
one-GPU box the N-device cases run with N = 1 or skip; nothing here needs more than the visible devices."""
import ctypes as C
import gzip
import json
import os
import subprocess
import sys

import numpy as np
import pytest

import gnumap_amd as g
from conftest import GOLDEN, ROOT

pytestmark = pytest.mark.gpu
EXE = os.path.join(ROOT, "gnumap_amd", "bin", "gnumap")
RUNS = os.path.join(GOLDEN, "ref_runs")


def n_devices():
    import torch
    return torch.cuda.device_count()            # counts devices without initialising the GPU in this process


def _track(text):
    d = {}
    for line in text.splitlines():
        f = line.split("\t")
        d[(f[0], int(f[1]))] = float(f[2])
    return d


@pytest.mark.parametrize("mode", ["default", "bs"])
def test_driver_binary_on_every_visible_gpu(mode, tmp_path):
    """gnumap --gpus N: the reads go to N x workers (gm_batch, stream) pairs on N index replicas; SAM = the reference program's record
    set (block order across GPUs is kept by the in-order writer: same text), track = the all-reduced sum"""
    n = n_devices()
    m = json.load(open(os.path.join(RUNS, "manifest.json")))[mode]
    out = str(tmp_path / "o")
    r = subprocess.run([EXE, "-g", os.path.join(GOLDEN, "syn.fa"), "-o", out, "-a", "0.9", f"--gpus={n}", "--chunk_reads=40", "--workers=2"] + m["argv"]
                       + [os.path.join(GOLDEN, m["fastq"])], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    assert f"{n} GPU(s)" in r.stderr
    sam = "".join(l for l in open(out + ".sam") if not l.startswith("@PG"))
    assert sam == gzip.open(os.path.join(RUNS, f"{mode}.sam.gz"), "rt").read()
    ext = m["tracks"][0]
    mine, ref = open(out + "." + ext).read(), gzip.open(os.path.join(RUNS, f"{mode}.{ext}.gz"), "rt").read()
    if ext == "sgr":
        a, b = _track(mine), _track(ref)
        for k in set(a) ^ set(b):
            assert (a.get(k) or b.get(k)) < 2e-3, k
        for k in set(a) & set(b):
            assert abs(a[k] - b[k]) <= 1e-4 * max(1.0, abs(b[k])) + 2e-5, (k, a[k], b[k])
    else:
        assert len(mine.splitlines()) == len(ref.splitlines())


def test_coverage_allreduce_over_all_devices(syn_fa):
    """gm_coverage_allreduce (ncclCommInitAll + in-place ncclAllReduce on every replica's HBM track): each device deposits its own
    pattern + a common one; afterwards EVERY replica holds the sum"""
    n = n_devices()
    if n < 2:
        pytest.skip("one visible GPU: the one-rank form of this call is test_rccl_allreduce_path_on_one_rank")
    ixs = [g.Index(syn_fa, device=d, flags=0) for d in range(n)]
    for d, ix in enumerate(ixs):
        ix.coverage_reset(8)
        ix.coverage_add([800, 8000 * (d + 1)], [64, 16], [d + 1.0, 0.5])
    arr = (C.c_void_p * n)(*[ix.h for ix in ixs])
    assert g.lib().gm_coverage_allreduce(arr, n) == 0, g.lib().gm_last_error()
    want = np.zeros(ixs[0].coverage_bins(), np.float32)
    want[100:108] = 8 * n * (n + 1) / 2.0
    for d in range(n):
        want[1000 * (d + 1):1000 * (d + 1) + 2] = 4.0
    for ix in ixs:
        np.testing.assert_array_equal(ix.coverage_download(), want)
        ix.close()


@pytest.mark.parametrize("extra", [[], ["--mode", "1"], ["--max-kmer-hits", "40", "--repeats"]], ids=["normal", "bisulfite", "repeats_h40"])
def test_bench_allreduce_leg_with_real_coverage(extra, tmp_path):
    """bench.py end to end at toy size with the process group forced on (GM_FORCE_DIST=1: the RCCL path with the ranks the box has):
    every rank deposits the coverage of one gm_map_batch + gm_output_batch block, the track (and with -b the 5 per-nucleotide tracks) is
    all-reduced in place, reduced totals = sum of the per-rank totals; the parity sample of the same run has no mismatch"""
    env = dict(os.environ, GM_FORCE_DIST="1", GM_BENCH_DIR=str(tmp_path), MASTER_ADDR="127.0.0.1", MASTER_PORT="29591")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--genome-mbp", "6", "--contigs", "3", "--reads", "40000", "--mer", "9", "--jump", "5",
                        "--steps", "1", "--cpu-seconds", "0", "--abi-reads", "16384", "--abi-block", "8192", "--parity-sample", "24"] + extra,
                       capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stderr[-2500:]
    j = json.loads(r.stdout.strip().splitlines()[-1])
    ar = j["coverage_allreduce"]
    assert ar and ar["checked"] and ar["in_place"]
    assert len(ar["tracks"]) == (2 if "--mode" in extra else 1)
    for got, want in zip(ar["reduced_totals"], ar["sum_of_rank_totals"]):
        assert want > 1000 and abs(got - want) <= 1e-5 * want
    assert j["parity_sample"]["n"] >= 20 and j["parity_sample"]["mismatches"] == 0
    assert j["abi"]["sam_records"] > 8000
    assert j["roofline"]["frac"] > 0 and j["cpu_baseline"] is None
