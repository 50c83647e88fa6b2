#!/usr/bin/env python3
"""Generate the committed golden fixtures.  Run in the build container only (needs oracle/_ref,
i.e. /root/reference):   python tests/golden/make_fixtures.py

Outputs (all small, all DATA — inputs and expected outputs, no reference source text):
  syn.fa, syn.fq                      synthetic genome (3 contigs, N run, planted repeats) + reads
  syn.fa.gnumap.{pac,ann,amb,bwt,sa}  index files written by the REFERENCE's own bwa_index (via oracle/_ref)
  ref_vectors.npz                     outputs of the reference functions on seeded inputs:
                                      occ, SA intervals, locate, windows, PWM, self score, NW score, traceback
"""
import os, sys, ctypes, json
import ctypes as C
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "tests"))
from reflib import RefLib   # noqa: E402

rng = np.random.default_rng(20261003)
ACGT = np.frombuffer(b"ACGT", dtype=np.uint8)


def rand_seq(n):
    return ACGT[rng.integers(0, 4, n)].copy()


def revcomp(b):
    comp = np.zeros(256, np.uint8)
    for a, c in zip(b"ACGTacgtNn", b"TGCAtgcaNn"):
        comp[a] = c
    return comp[b[::-1]]


def make_genome():
    c1 = rand_seq(150_000)
    c2 = rand_seq(100_000)
    c3 = rand_seq(30_000)
    # planted 300-bp repeat on two contigs (multi-position ScoredSeq, X0:i:2) and a reverse-complement copy
    rep = c1[40_000:40_300].copy()
    c2[60_000:60_300] = rep
    c3[10_000:10_300] = revcomp(rep)
    # a second repeat family with 2 % divergence (different keys, shared votes)
    rep2 = c1[90_000:90_400].copy()
    mut = rep2.copy()
    idx = rng.choice(400, 8, replace=False)
    mut[idx] = ACGT[(np.searchsorted(ACGT, mut[idx]) + 1) % 4]
    c2[20_000:20_400] = mut
    # N run (exercises .amb + lrand48 fill) and some lowercase
    c1[120_000:120_150] = ord("N")
    c2[5_000:5_400] = np.char.lower(c2[5_000:5_400].view("S1")).view(np.uint8)
    # low-complexity stretch: many k-mer hits, exercises -h / -T
    c3[20_000:20_600] = np.tile(np.frombuffer(b"ACACACGT", np.uint8), 75)
    return [("chrA", "synthetic contig one", c1), ("chrB", "", c2), ("chrC", "third", c3)]


def write_fasta(path, contigs):
    with open(path, "wb") as f:
        for name, anno, seq in contigs:
            f.write(b">" + name.encode() + ((b" " + anno.encode()) if anno else b"") + b"\n")
            for i in range(0, len(seq), 70):
                f.write(seq[i:i + 70].tobytes() + b"\n")


def mutate(read, sub=0.01, ins=0.004, dele=0.004):
    out = []
    i = 0
    while i < len(read):
        r = rng.random()
        if r < dele:
            i += 1
            continue
        if r < dele + ins:
            out.append(ACGT[rng.integers(0, 4)])
        b = read[i]
        if rng.random() < sub:
            b = ACGT[(np.searchsorted(ACGT, np.uint8(chr(b).upper().encode()[0])) + rng.integers(1, 4)) % 4]
        out.append(b)
        i += 1
    return np.array(out, np.uint8)


def make_reads(contigs):
    recs = []
    offs = np.cumsum([0] + [len(c[2]) for c in contigs])
    whole = np.concatenate([c[2] for c in contigs])

    def add(name, seq, qual=None):
        if qual is None:
            qual = (33 + rng.integers(20, 41, len(seq))).astype(np.uint8)
        recs.append((name, bytes(seq), bytes(qual)))

    rid = 0
    for L, n in ((100, 360), (50, 60), (150, 40), (36, 20)):
        for _ in range(n):
            ci = rng.integers(0, len(contigs))
            clen = len(contigs[ci][2])
            p = rng.integers(0, clen - L - 8)
            frag = contigs[ci][2][p:p + L + 8]
            frag = np.array([ord(chr(x).upper()) for x in frag], np.uint8)
            frag[frag == ord("N")] = ord("A")
            r = mutate(frag)[:L]
            if len(r) < L:
                continue
            strand = rng.integers(0, 2)
            if strand:
                r = revcomp(r)
            add(f"r{rid}_{contigs[ci][0]}_{p}_{'-' if strand else '+'}_{L}", r)
            rid += 1
    # reads inside the planted repeats (multi-position hits), both strands
    for k in range(24):
        p = 40_000 + rng.integers(0, 200)
        r = contigs[0][2][p:p + 100].copy()
        if k % 2:
            r = revcomp(r)
        add(f"rep{k}_{p}", r)
    for k in range(12):
        p = 90_000 + rng.integers(0, 300)
        add(f"repdiv{k}_{p}", contigs[0][2][p:p + 100].copy())
    # low complexity reads (many hits)
    for k in range(8):
        p = 20_000 + rng.integers(0, 480)
        add(f"lowc{k}_{p}", contigs[2][2][p:p + 100].copy())
    # reads spanning a contig end / the genome end
    for k, gpos in enumerate((offs[1] - 60, offs[1] - 30, offs[2] - 50, offs[3] - 100, offs[3] - 99)):
        r = whole[gpos:gpos + 100].copy()
        r[r == ord("N")] = ord("A")
        add(f"edge{k}_{gpos}", np.array([ord(chr(x).upper()) for x in r], np.uint8))
    # reads at the very start of the genome (beginning clamps to 0)
    add("start0", whole[0:100].copy())
    add("start3", whole[3:103].copy())
    add("start0rc", revcomp(whole[0:100].copy()))
    # reads containing N, lowercase, low quality, random (unmappable), too short
    r = whole[7_000:7_100].copy(); r[10] = ord("N"); r[55] = ord("N"); add("withN", r)
    r = whole[8_000:8_100].copy(); add("lower", np.char.lower(r.view("S1")).view(np.uint8))
    r = whole[9_000:9_100].copy(); add("lowq", r, (33 + rng.integers(2, 8, 100)).astype(np.uint8))
    r = whole[9_500:9_600].copy(); add("q0", r, np.full(100, 33, np.uint8))
    r = whole[9_700:9_800].copy(); add("qmax", r, np.full(100, 33 + 41, np.uint8))
    for k in range(10):
        add(f"random{k}", rand_seq(100))
    add("short8", whole[100:108].copy())
    add("len10", whole[200:210].copy())
    add("len11", whole[300:311].copy())
    # reads over the N run (reference bases there were replaced by lrand48()&3)
    r = whole[119_950:120_050].copy(); r[r == ord("N")] = ord("C"); add("overN", r)
    return recs


def write_fastq(path, recs):
    with open(path, "wb") as f:
        for name, seq, qual in recs:
            f.write(b"@" + name.encode() + b"\n" + seq + b"\n+\n" + qual + b"\n")


def main():
    fa = os.path.join(HERE, "syn.fa")
    fq = os.path.join(HERE, "syn.fq")
    contigs = make_genome()
    write_fasta(fa, contigs)
    recs = make_reads(contigs)
    write_fastq(fq, recs)
    ref = RefLib()
    for ext in ("pac", "ann", "amb", "bwt", "sa"):
        p = f"{fa}.gnumap.{ext}"
        if os.path.exists(p):
            os.remove(p)
    assert ref.index_build(fa) == 0
    ref.setup(0)
    ix = ref.index_load(fa)
    vec = {}
    seq_len = ref.lib.ref_seq_len(ix)
    l_pac = seq_len
    vec["seq_len"] = np.uint64(seq_len)
    vec["primary"] = np.uint64(ref.lib.ref_primary(ix))
    # --- occ ---
    ks = np.concatenate([rng.integers(0, seq_len + 1, 2000).astype(np.uint64),
                         np.array([0, 1, 127, 128, seq_len - 1, seq_len, 2**64 - 1, int(vec["primary"]), int(vec["primary"]) - 1, int(vec["primary"]) + 1], np.uint64)])
    cs = rng.integers(0, 4, len(ks)).astype(np.int32)
    vec["occ_k"] = ks; vec["occ_c"] = cs
    vec["occ_out"] = np.array([ref.lib.ref_occ(ix, int(k), int(c)) for k, c in zip(ks, cs)], np.uint64)
    # --- SA intervals: k-mers from reads (both present and absent), several lengths ---
    whole = np.concatenate([c[2] for c in contigs])
    kmers = []
    for m in (6, 10, 12, 16, 20, 32):
        for _ in range(150):
            p = rng.integers(0, l_pac - m)
            k = whole[p:p + m].copy()
            if rng.random() < 0.3:
                k[rng.integers(0, m)] = ACGT[rng.integers(0, 4)]
            if rng.random() < 0.05:
                k[rng.integers(0, m)] = ord("N")
            if rng.random() < 0.2:
                k = np.char.lower(k.view("S1")).view(np.uint8)
            kmers.append(bytes(k))
    iv = np.zeros((len(kmers), 2), np.uint64)
    for i, k in enumerate(kmers):
        iv[i] = ref.sa_interval(ix, k)
    vec["kmers"] = np.array(kmers, dtype="S32"); vec["kmer_iv"] = iv
    # --- locate ---
    ranks = np.concatenate([rng.integers(1, seq_len + 1, 3000).astype(np.uint64), np.array([1, 2, 31, 32, 33, seq_len], np.uint64)])
    vec["loc_rank"] = ranks
    vec["loc_out"] = np.array([ref.lib.ref_sa_coord(ix, int(r)) for r in ranks], np.uint64)
    # --- windows (incl. contig boundaries and genome end) ---
    offs = np.cumsum([0] + [len(c[2]) for c in contigs])
    begins = list(rng.integers(0, l_pac - 1, 300)) + [0, 1, offs[1] - 100, offs[1] - 99, offs[1] - 1, offs[1], offs[2] - 50, offs[3] - 100, offs[3] - 99, offs[3] - 1, 119_990]
    wl = [int(x) for x in rng.choice([36, 50, 100, 150], len(begins))]
    wins = []
    for b, L in zip(begins, wl):
        wins.append(ref.window(ix, int(b), L))
    vec["win_begin"] = np.array(begins, np.uint64); vec["win_len"] = np.array(wl, np.int32)
    vec["win_out"] = np.array(wins, dtype="S160")
    # --- FASTQ -> PWM through the reference's SeqReader, self score, NW score, traceback ---
    n, lens, pwm, seqs, fqs, names = ref.read_fastq(fq, 0, 700, 160)
    assert n == len(recs), (n, len(recs))
    vec["fq_len"] = lens[:n]; vec["fq_pwm"] = pwm[:n]
    self_scores = np.zeros(n, np.float32)
    for i in range(n):
        self_scores[i] = ref.self_score(pwm[i, :lens[i]], seqs[i])
    vec["self_score"] = self_scores
    # NW + traceback cases: read i vs (true window with jitter -3..3 | random window)
    cases = []
    for i in range(n):
        L = int(lens[i])
        if L < 20:
            continue
        nm = names[i]
        parts = nm.split("_")
        for t in range(3):
            b = int(rng.integers(0, l_pac - L))
            if parts[0].startswith("r") and len(parts) == 5 and t < 2:
                ci = [c[0] for c in contigs].index(parts[1])
                b = int(offs[ci]) + int(parts[2]) + int(rng.integers(-3, 4))
                b = max(0, min(b, int(l_pac) - L))
            w = ref.window(ix, b, L)
            if not w:
                continue
            rc = 1 if (len(parts) == 5 and parts[3] == "-") else 0
            cases.append((i, b, rc, w))
    cases = cases[:1500]
    nw_scores = np.zeros(len(cases), np.float32)
    tb_aligned = []; tb_cigar = []; tb_len = np.zeros(len(cases), np.int32)
    from reflib import revcomp_pwm, revcomp_str
    for ci, (i, b, rc, w) in enumerate(cases):
        L = int(lens[i])
        P = pwm[i, :L]
        cons = seqs[i]
        if rc:
            P = revcomp_pwm(P); cons = revcomp_str(cons)
        nw_scores[ci] = ref.nw_score(P, w)
        al, alen, cg = ref.traceback(P, cons, w)
        tb_aligned.append(al); tb_cigar.append(cg); tb_len[ci] = alen
    vec["nw_read"] = np.array([c[0] for c in cases], np.int32)
    vec["nw_begin"] = np.array([c[1] for c in cases], np.uint64)
    vec["nw_rc"] = np.array([c[2] for c in cases], np.int8)
    vec["nw_window"] = np.array([c[3] for c in cases], dtype="S160")
    vec["nw_score"] = nw_scores
    vec["tb_aligned_hex"] = np.array([a.hex() for a in tb_aligned], dtype="S700")
    vec["tb_len"] = tb_len
    vec["tb_cigar"] = np.array(tb_cigar, dtype="S256")
    # output-side helpers of inc/SequenceOperations.h: reverse_comp (:56-96), reverse_CIGAR (:109-123), fix_CIGAR_for_deletions (:32-42)
    helper_cigars = sorted(set(tb_cigar)) + [b"97M3D", b"3D97M", b"100M", b"5M2D", b"1D", b"12M1I3D", b"10M:5M", b"7"]
    out = C.create_string_buffer(4096)
    fixc = []; revc = []
    for cg in helper_cigars:
        ref.lib.ref_fix_cigar(cg, out); fixc.append(out.value)
        ref.lib.ref_reverse_cigar(cg, out); revc.append(out.value)
    vec["hc_in"] = np.array(helper_cigars, dtype="S256"); vec["hc_fix"] = np.array(fixc, dtype="S256"); vec["hc_rev"] = np.array(revc, dtype="S256")
    helper_strs = [seqs[i] for i in range(0, n, 7)] + [a for a in tb_aligned[:200]] + [b"acgtnACGTN-xyzRYKM", b"", b"a", b"-"]
    helper_strs = [x for x in helper_strs if b"\0" not in x]
    rcs = []
    for st in helper_strs:
        ref.lib.ref_reverse_comp(st, out); rcs.append(out.value)
    vec["rc_in_hex"] = np.array([x.hex() for x in helper_strs], dtype="S4200"); vec["rc_out_hex"] = np.array([x.hex() for x in rcs], dtype="S4200")
    table, gap, maxgap = ref.get_scores()
    vec["S"] = table; vec["gap"] = np.float32(gap); vec["max_gap"] = np.int32(maxgap)
    np.savez_compressed(os.path.join(HERE, "ref_vectors.npz"), **vec)
    print("fixtures written:", n, "reads,", len(cases), "NW cases,", len(kmers), "k-mers")


if __name__ == "__main__":
    main()
