import sys, os, json, numpy as np
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import gnumap_amd as g
from conftest import read_fastq
reads = read_fastq(os.path.join(ROOT, 'tests', 'golden', 'syn.fq'))
ix = g.Index(os.path.join(ROOT, 'tests', 'golden', 'syn.fa'), flags=g.GM_INDEX_FULL_SA)
p = g.Params()
for r in (529, 531, 530):
    B, Q, Ln = g.pack_reads([reads[r][1]], [reads[r][2]])
    b = g.Batch(ix, 1, B.shape[1])
    b.upload(p, B, Q, Ln); b.map_device(p)
    hits, status, self_score, top = b.raw_hits()
    c = b.counters()
    print(r, reads[r][0], len(reads[r][1]), hits.tolist(), {k: c[k] for k in ("seeds_used", "sa_hits", "candidates", "vote_retries")})
