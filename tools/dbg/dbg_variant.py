import sys, os, json, numpy as np
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import gnumap_amd as g
from conftest import read_fastq
reads = read_fastq(os.path.join(ROOT, 'tests', 'golden', 'syn.fq'))
B, Q, Ln = g.pack_reads([r[1] for r in reads], [r[2] for r in reads])
ix = g.Index(os.path.join(ROOT, 'tests', 'golden', 'syn.fa'), flags=g.GM_INDEX_FULL_SA)
p = g.Params()
b = g.Batch(ix, len(reads), B.shape[1])
b.upload(p, B, Q, Ln); b.map_device(p)
hits, status, self_score, top = b.raw_hits()
np.save(sys.argv[1], hits)
print(len(hits), b.counters())
