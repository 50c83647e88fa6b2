#!/usr/bin/env python3
"""Real-sequence fixture = BASELINE configs[0] (examples/Cel_gen.* of the reference).  BUILD CONTAINER ONLY: needs /root/reference and
oracle/_ref/gnumap_ref (the unmodified reference program, oracle/Makefile refbin).

The reference's examples hold the READS of its C. elegans example (Cel_gen.reads.fq: 24 869 ART-simulated 50-bp reads with ART's
Illumina quality profile, Phred down to ~6) and ART's truth alignment (Cel_gen.reads.aln: contig chrI_third, 4 973 850 bp; for every read
its start, strand and the REFERENCE bases under it) - but not the genome (Cel_gen.fa is listed in .MISSING_LARGE_BLOBS).  This script
rebuilds the genome as far as the data holds it: the .aln reference strings are laid down at their positions (overlapping reads must
agree - checked), every base no read covers comes from a seeded generator.  The result is a real-sequence reference under every read
(real C. elegans low-complexity / repeats where the reads fall) with real quality strings.

Then the reference PROGRAM runs on genome + Cel_gen.reads.fq (README.md:22's command with the full read file) in three modes and its
outputs are committed as the expected values:
    default (-a 0.9)  -> default.sam.gz, default.sgr.gz
    --no_nw           -> no_nw.sam.gz
    -b                -> bs.sam.gz, bs.gmp.gz
plus truth.npz (forward-strand start + strand of every read, from the .aln) and the reads themselves (reads.fq.gz: a data file of the
reference's examples, unchanged).

ART .aln: '>ref  read_id  aln_start  strand', then the reference line and the read line ('-' = gap).  aln_start is 0-based and relative
to the strand the read came from: a '-' read at start s covering n reference bases lies at forward position G - s - n, and its reference
line is the reverse complement of the forward bases there.
"""
import gzip
import os
import shutil
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(HERE)))
REF = "/root/reference"
EXE = os.path.join(ROOT, "oracle", "_ref", "gnumap_ref")
COMP = bytes.maketrans(b"ACGTacgt", b"TGCAtgca")


def parse_aln(path):
    G = None; name = None; recs = []
    with open(path, "rb") as f:
        lines = f.read().split(b"\n")
    i = 0
    while not lines[i].startswith(b"##Header End"):
        if lines[i].startswith(b"@SQ"):
            _, name, G = lines[i].split(b"\t"); G = int(G)
        i += 1
    i += 1
    while i + 2 < len(lines) + 1 and lines[i].startswith(b">"):
        f = lines[i][1:].split(b"\t")
        recs.append((f[1].decode(), int(f[2]), f[3].decode(), lines[i + 1], lines[i + 2]))
        i += 3
    return name.decode(), G, recs


def main():
    name, G, recs = parse_aln(os.path.join(REF, "examples", "Cel_gen.reads.aln"))
    rng = np.random.default_rng(20240101)
    genome = np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, G)].copy()
    known = np.zeros(G, bool)
    truth = {}
    clashes = 0
    for rid, start, strand, refline, readline in recs:
        ref = refline.replace(b"-", b"")
        n = len(ref)
        if strand == "+":
            pos, fwd = start, ref
        else:
            pos, fwd = G - start - n, ref.translate(COMP)[::-1]
        assert 0 <= pos and pos + n <= G, (rid, pos, n)
        seg = np.frombuffer(fwd, np.uint8)
        k = known[pos:pos + n]
        clashes += int((genome[pos:pos + n][k] != seg[k]).sum())
        genome[pos:pos + n] = seg
        known[pos:pos + n] = True
        truth[rid] = (pos, 0 if strand == "+" else 1, n)
    # overlapping reads were cut from ONE genome: any disagreement means the format was misread
    assert clashes == 0, f"{clashes} bases disagree between overlapping .aln records"
    print(f"{name}: {G} bp, {len(recs)} truth records, {int(known.sum())} bases ({known.mean() * 100:.1f} %) from the .aln, the rest seeded random")

    fa = os.path.join(HERE, "celgen.fa")
    with open(fa, "wb") as f:
        f.write(b">" + name.encode() + b"\n")
        for s in range(0, G, 70):
            f.write(genome[s:s + 70].tobytes() + b"\n")
    fq_src = os.path.join(REF, "examples", "Cel_gen.reads.fq")
    names = [l[1:].strip().decode() for i, l in enumerate(open(fq_src, "rb")) if i % 4 == 0]
    assert len(names) == len(recs) and set(names) == set(truth)
    np.savez_compressed(os.path.join(HERE, "truth.npz"), pos=np.array([truth[n][0] for n in names], np.int64),
                        strand=np.array([truth[n][1] for n in names], np.int8), span=np.array([truth[n][2] for n in names], np.int16))
    with open(fq_src, "rb") as f, gzip.GzipFile(os.path.join(HERE, "reads.fq.gz"), "wb", 9, mtime=0) as z:
        z.write(f.read())

    wd = tempfile.mkdtemp(prefix="celgen_")
    try:
        wfa = os.path.join(wd, "celgen.fa"); shutil.copy(fa, wfa)
        wfq = os.path.join(wd, "reads.fq"); shutil.copy(fq_src, wfq)
        # a first run builds the index files (and is thrown away): the reference never zeroes amount_genome (src/GenomeBwt.cpp:483-490 relies on
        # fresh pages), and in the run that ALSO built the index the array lands in recycled heap memory - garbage bins in its .sgr
        one = os.path.join(wd, "one.fq")
        with open(wfq, "rb") as f, open(one, "wb") as o:
            o.write(b"".join(f.readline() for _ in range(4)))
        r = subprocess.run([EXE, "-g", wfa, "-o", os.path.join(wd, "warm"), "-a", "0.9", one], capture_output=True, text=True)
        assert r.returncode == 0, r.stderr[-2000:]
        for mode, flags, tracks in (("default", [], ["sgr"]), ("no_nw", ["--no_nw"], []), ("bs", ["-b"], ["gmp"])):
            out = os.path.join(wd, mode)
            r = subprocess.run([EXE, "-g", wfa, "-o", out, "-a", "0.9", "-c", "1"] + flags + [wfq], capture_output=True, text=True)
            assert r.returncode == 0, r.stderr[-2000:]
            sam = [l for l in open(out + ".sam", "rb") if not l.startswith(b"@PG")]
            with gzip.GzipFile(os.path.join(HERE, f"{mode}.sam.gz"), "wb", 9, mtime=0) as z:
                z.write(b"".join(sam))
            for t in tracks:
                with open(out + "." + t, "rb") as f, gzip.GzipFile(os.path.join(HERE, f"{mode}.{t}.gz"), "wb", 9, mtime=0) as z:
                    z.write(f.read())
            print(mode, sum(1 for l in sam if not l.startswith(b"@")), "SAM records")
            if "sgr" in tracks:            # a clean track: no bin can hold more than bin_size x (reads that could overlap it)
                worst = max(float(l.split(b"\t")[2]) for l in open(out + ".sgr", "rb"))
                assert worst < 8 * 1000, f"garbage in the reference's .sgr (max bin {worst}): amount_genome was not fresh memory"
    finally:
        shutil.rmtree(wd, ignore_errors=True)
    subprocess.run(["gzip", "-9", "-n", "-f", fa], check=True)


if __name__ == "__main__":
    sys.exit(main())
