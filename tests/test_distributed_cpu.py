"""The N > 1 path on CPU: two gloo ranks shard the fixture reads, map their shard (with the oracle standing in for the
device, which CPU tests cannot use), all-reduce the coverage track with the same helper bench.py uses, and must reproduce the
single-process result: the union of the per-rank SAM shards equals the single run, and the reduced track equals the single
track (the reference's MPI read-sharding mode, src/Driver.cpp:1617-1811)."""
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r'''
import os, sys
import numpy as np, torch
sys.path.insert(0, {root!r}); sys.path.insert(0, os.path.join({root!r}, "tests"))
from gnumap_amd import dist as gd
from reflib import OracleLib
rank, world = gd.init("gloo")
out = sys.argv[1]
fq = os.path.join({root!r}, "tests", "golden", "syn.fq"); fa = os.path.join({root!r}, "tests", "golden", "syn.fa")
lines = open(fq, "rb").read().split(b"\n")
n = len(lines) // 4
lo, hi = gd.shard_range(n, rank, world)
shard = os.path.join(out, f"shard{{rank}}.fq")
open(shard, "wb").write(b"\n".join(lines[4 * lo:4 * hi]) + b"\n")
orc = OracleLib(); ix = orc.index_load(fa); p = orc.params()
st = orc.run(ix, p, shard, os.path.join(out, f"out_{{rank}}"), threads=2)
# per-rank coverage track from the .sgr the oracle wrote (bins of 8 on the concatenated coordinate)
offs = {{"chrA": 0, "chrB": 150000, "chrC": 250000}}
track = torch.zeros(280000 // 8 + 64, dtype=torch.float32)
for l in open(os.path.join(out, f"out_{{rank}}.sgr")):
    c, pos, v = l.split("\t"); track[(offs[c] + int(pos) - 1) // 8] = float(v)
gd.allreduce_coverage(track)
tmax = gd.max_over_ranks(0.5 + rank)
total = gd.sum_over_ranks(st.n_reads)
gd.barrier()
if rank == 0:
    np.save(os.path.join(out, "track.npy"), track.numpy())
    open(os.path.join(out, "meta.txt"), "w").write(f"{{tmax}} {{total}} {{world}}")
'''


@pytest.mark.timeout(300)
def test_two_gloo_ranks_equal_single_process(tmp_path, oracle, syn_fa, syn_fq):
    out = str(tmp_path)
    script = os.path.join(out, "worker.py")
    open(script, "w").write(WORKER.format(root=ROOT))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29577", OMP_NUM_THREADS="1")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", "29577", script, out], capture_output=True, text=True, env=env, timeout=280)
    assert r.returncode == 0, r.stderr[-2000:]
    tmax, total, world = open(os.path.join(out, "meta.txt")).read().split()
    assert float(tmax) == 1.5 and int(float(total)) == 551 and int(world) == 2
    # single-process truth
    ix = oracle.index_load(syn_fa)
    oracle.run(ix, oracle.params(), syn_fq, os.path.join(out, "single"), threads=2)
    body = lambda f: [l for l in open(f) if not l.startswith("@")]
    merged = body(os.path.join(out, "out_0.sam")) + body(os.path.join(out, "out_1.sam"))      # contiguous shards: concatenation keeps read order
    assert merged == body(os.path.join(out, "single.sam"))
    offs = {"chrA": 0, "chrB": 150000, "chrC": 250000}
    single = np.zeros(280000 // 8 + 64, np.float32)
    for l in open(os.path.join(out, "single.sgr")):
        c, pos, v = l.split("\t"); single[(offs[c] + int(pos) - 1) // 8] = float(v)
    track = np.load(os.path.join(out, "track.npy"))
    # a bin that only reaches the 0.001 print threshold after the reduction is missing from the per-rank files: compare above it
    big = single > 0.01
    np.testing.assert_allclose(track[big], single[big], rtol=1e-4, atol=2e-3)


def test_shard_range_covers_everything():
    from gnumap_amd import dist as gd
    for n in (0, 1, 7, 551, 10_000_001):
        for world in (1, 2, 3, 8):
            got = []
            for r in range(world):
                lo, hi = gd.shard_range(n, r, world)
                assert 0 <= lo <= hi <= n
                got += list(range(lo, hi)) if n < 1000 else [(lo, hi)]
            if n < 1000:
                assert got == list(range(n))
            else:
                assert got[0][0] == 0 and got[-1][1] == n and all(a[1] == b[0] for a, b in zip(got, got[1:]))
