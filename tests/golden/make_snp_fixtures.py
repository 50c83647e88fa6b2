#!/usr/bin/env python3
"""BUILD CONTAINER ONLY.  Function-level pin of the --snp deposit (SNPScoredSeq::score, src/SNPScoredSeq.cpp:25-109): the UNMODIFIED
bin_seq::pairHMM (src/bin_seq.cpp:60-244, through oracle/ref_harness.cpp::ref_pair_hmm) on reads of the fixture against reference
windows - true loci (both strands, reads with indels), windows shifted by up to 3 bases, an unrelated window, reads containing N and
low qualities - and its 5 floats per window position stored bit for bit -> tests/golden/ref_vectors_snp.npz.

The reference PROGRAM cannot pin this mode end to end here: its --snp run aborts in PrintFinalSNP on gsl_cdf_chisq_P (GSL is not in the
image; oracle/Makefile leaves that one symbol unresolved), after the SAM file - which equals the default mode's - has been written."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, ".."))
sys.path.insert(0, os.path.join(HERE, "..", ".."))
from conftest import read_fastq          # noqa: E402
from reflib import OracleLib, RefLib, revcomp_pwm   # noqa: E402


def argmax_cons(P):
    out = bytearray()
    for c in P:
        if c[0] == c[1] == c[2] == c[3]:
            out.append(ord("n"))
        elif c[0] >= c[1]:
            out.append(ord("a" if c[0] >= c[3] else "t") if c[0] >= c[2] else ord("g" if c[2] >= c[3] else "t"))
        else:
            out.append(ord("c" if c[1] >= c[3] else "t") if c[1] >= c[2] else ord("g" if c[2] >= c[3] else "t"))
    return bytes(out)


def main():
    ref = RefLib(); ref.setup(0)
    orc = OracleLib()
    fa = os.path.join(HERE, "syn.fa")
    oix = orc.index_load(fa)
    op = orc.params()
    reads = read_fastq(os.path.join(HERE, "syn.fq"))
    rng = np.random.default_rng(7)
    rows = []
    for k, (name, seq, qual) in enumerate(reads):
        if len(rows) >= 90:
            break
        if len(seq) < 20:
            continue
        P = orc.pwm(seq, qual)
        o = orc.map_read(oix, op, P, seq)
        if not o["hits"]:
            continue
        h = o["hits"][0]
        pos, strand = h["pos"][0]
        L = len(seq)
        for shift in ((0,) if k % 3 else (0, int(rng.integers(-3, 4)), 777)):
            b = int(pos) + shift
            w = orc.window(oix, b, L)
            if len(w) != L:
                continue
            Pq = revcomp_pwm(P) if strand else P
            cons = argmax_cons(Pq)
            rows.append((k, strand, b, np.ascontiguousarray(Pq, np.float32), cons, w, ref.pair_hmm(Pq, cons, w)))
    n = len(rows); Lmax = max(len(r[5]) for r in rows)
    pwm = np.zeros((n, Lmax, 4), np.float32); out = np.zeros((n, Lmax, 5), np.float32); lens = np.zeros(n, np.int32)
    cons = np.zeros((n, Lmax), np.uint8); win = np.zeros((n, Lmax), np.uint8)
    for i, (k, strand, b, P, c, w, o) in enumerate(rows):
        L = len(w); lens[i] = L; pwm[i, :L] = P; out[i, :L] = o; cons[i, :L] = np.frombuffer(c, np.uint8); win[i, :L] = np.frombuffer(w, np.uint8)
    np.savez_compressed(os.path.join(HERE, "ref_vectors_snp.npz"), read=np.array([r[0] for r in rows], np.int32), strand=np.array([r[1] for r in rows], np.int8),
                        begin=np.array([r[2] for r in rows], np.int64), len=lens, pwm=pwm, cons=cons, window=win, hmm=out)
    print(n, "pair-HMM vectors;", sum(1 for r in rows if r[1]), "on the reverse strand; row sums", float(out[0, :lens[0]].sum(1).min()), float(out[0, :lens[0]].sum(1).max()))


if __name__ == "__main__":
    main()
