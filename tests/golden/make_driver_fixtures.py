#!/usr/bin/env python3
"""Driver-level golden fixtures, produced by the UNMODIFIED reference program.

Run in the build container only (needs /root/reference):
    make -C oracle refbin && python tests/golden/make_driver_fixtures.py

oracle/_ref/gnumap_ref is src/Driver.cpp (+ GenomeBwt, Genome, the ScoredSeq family, bin_seq, SeqReader, the vendored BWA
files) compiled in place by oracle/Makefile.  This script runs it on the committed syn.fa / syn.fq (+ syn_ill.fq, Phred+64
qualities with a mid-file fallback) once per entry of MODES and commits what it wrote:

    tests/golden/ref_runs/<mode>.sam.gz        the SAM file as written with -c 1 (record order included), @PG line dropped
    tests/golden/ref_runs/<mode>.sgr.gz|.gmp.gz  the coverage track / per-nucleotide track text
    tests/golden/ref_runs/manifest.json        argv per mode + the oracle / product parameter names they correspond to

Everything committed is DATA (outputs of the reference on our inputs); no reference source text.
--print_all_sam modes expose every ScoredSeq of every read (score XA, posterior XP = exp(score)/denominator, position
set X0 + rows), i.e. the content of gReadLocs / gReadDenominator, through the reference's own writer.
"""
import gzip, json, os, subprocess, sys, tempfile, shutil
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REFBIN = os.path.join(ROOT, "oracle", "_ref", "gnumap_ref")
OUT = os.path.join(HERE, "ref_runs")

# name -> (reference argv between "-a 0.9" and the FASTQ, oracle/product parameter overrides, fastq)
MODES = {
    "default":        ([], {}, "syn.fq"),
    "no_nw":          (["--no_nw"], dict(nw=0), "syn.fq"),
    "bs":             (["-b"], dict(mode=1), "syn.fq"),
    "b2":             (["--b2"], dict(mode=2), "syn.fq"),
    "atog":           (["-d"], dict(mode=3), "syn.fq"),
    "h30":            (["-h", "30"], dict(max_kmer_hits=30), "syn.fq"),
    "T2":             (["-T", "2"], dict(max_matches=2), "syn.fq"),
    "u":              (["-u", "1"], dict(unique_only=1), "syn.fq"),            # -u swallows the next argv (Driver.cpp:2768-2770)
    "all":            (["--print_all_sam"], dict(print_all_sam=1), "syn.fq"),
    "m16":            (["-m", "16"], dict(mer=16), "syn.fq"),
    "m16_h150_all":   (["-m", "16", "-h", "150", "--print_all_sam"], dict(mer=16, max_kmer_hits=150, print_all_sam=1), "syn.fq"),
    "k3":             (["-k", "3"], dict(min_seed_hits=3), "syn.fq"),
    "k1_all":         (["-k", "1", "--print_all_sam"], dict(min_seed_hits=1, print_all_sam=1), "syn.fq"),
    "m6_j2":          (["-m", "6", "-j", "2"], dict(mer=6, jump=2), "syn.fq"),
    "m20_j2":         (["-m", "20", "-j", "2"], dict(mer=20, jump=2), "syn.fq"),
    "m14_j7_all":     (["-m", "14", "-j", "7", "--print_all_sam"], dict(mer=14, jump=7, print_all_sam=1), "syn.fq"),
    "a95":            (["-a", "0.95"], dict(align_score=0.95), "syn.fq"),
    "a80_all":        (["-a", "0.8", "--print_all_sam"], dict(align_score=0.8, print_all_sam=1), "syn.fq"),
    "q60":            (["-q", "60"], dict(cutoff=60.0), "syn.fq"),
    "raw60":          (["-r", "-a", "60"], dict(align_is_fraction=0, align_score=60.0), "syn.fq"),
    "G6":             (["-G", "-6"], dict(gap=-6.0), "syn.fq"),
    "fast":           (["--fast"], dict(fast=1, mer=14), "syn.fq"),
    "up":             (["--up_strand"], dict(neg_strand=0), "syn.fq"),
    "down":           (["--down_strand"], dict(pos_strand=0), "syn.fq"),
    "bin1":           (["--bin_size=1"], dict(bin_size=1), "syn.fq"),
    "bs_all":         (["-b", "--print_all_sam"], dict(mode=1, print_all_sam=1), "syn.fq"),
    "no_nw_k3_all":   (["--no_nw", "-k", "3", "--print_all_sam"], dict(nw=0, min_seed_hits=3, print_all_sam=1), "syn.fq"),
    "illumina_fallback_first": (["--illumina"], dict(illumina=1), "syn.fq"),   # first read already shows Q < '@': Phred+33 throughout
    "illumina":       (["--illumina"], dict(illumina=1), "syn_ill.fq"),        # Phred+64 reads, then a fallback in mid-file
    "illumina_all":   (["--illumina", "--print_all_sam"], dict(illumina=1, print_all_sam=1), "syn_ill.fq"),
    # -S: 5 x 4 substitution file (readPWM Driver.cpp:768-859); "_subst" is not a parameter field: the tests apply the file to the
    # finalized parameters (OracleLib.apply_subst / gm_params_load_subst)
    "subst_all":      (["-S", "subst.txt", "--print_all_sam"], dict(print_all_sam=1, _subst="subst.txt"), "syn.fq"),
    "subst_bs":       (["-S", "subst.txt", "-b"], dict(mode=1, _subst="subst.txt"), "syn.fq"),
    # -M: band half-width of the DP (gMAX_GAP); the product runs these on the generic band kernels (gm_band.hip)
    "M1":             (["-M", "1"], dict(max_gap=1), "syn.fq"),
    "M2_all":         (["-M", "2", "--print_all_sam"], dict(max_gap=2, print_all_sam=1), "syn.fq"),
    "M5_all":         (["-M", "5", "-a", "0.8", "--print_all_sam"], dict(max_gap=5, align_score=0.8, print_all_sam=1), "syn.fq"),
    "M7_bs":          (["--max_gap=7", "-b"], dict(max_gap=7, mode=1), "syn.fq"),
    "M4_ill":         (["-M", "4", "--illumina"], dict(max_gap=4, illumina=1), "syn_ill.fq"),
    # malformed FASTQ records: the reader shifts lines until it is back in step and goes on (SeqReader.cpp:1091-1144)
    "malformed":      ([], {}, "syn_bad.fq"),
    "malformed_tail": (["--no_nw"], dict(nw=0), "syn_bad2.fq"),               # ... and an input that ends inside a record, without a final newline
}


def make_illumina_fastq(src, dst):
    """90 reads of syn.fq re-encoded Phred+64; read 60 keeps one Phred+33 character -> the reference switches
    --illumina off there and reads 60.. are taken as Phred+33 (their Phred+64 characters then mean Q 31..71)."""
    recs = []
    with open(src, "rb") as f:
        while True:
            name = f.readline()
            if not name:
                break
            seq = f.readline(); f.readline(); qual = f.readline().rstrip(b"\n")
            recs.append((name, seq, qual))
    rng = np.random.default_rng(7)
    pick = [recs[i] for i in sorted(rng.choice(420, 90, replace=False))]
    with open(dst, "wb") as f:
        for i, (name, seq, qual) in enumerate(pick):
            q = bytearray(min(126, c + 31) for c in qual)
            if i == 60:
                q[5] = ord("5")
            f.write(name + seq + b"+\n" + bytes(q) + b"\n")


def make_malformed_fastq(src, dst, dst2):
    """syn_bad.fq: 150 reads of syn.fq with broken records in between - a '+' line that is something else, a quality line shorter than
    its sequence, a name line without '@', a stray line between two records, blank lines, a record cut after its sequence, a quality
    line LONGER than its sequence (accepted as it is), a quality line starting with '@'.  syn_bad2.fq: 40 reads, a broken record near
    the end, the last record cut inside its quality line's predecessor and no newline at the end of the file."""
    recs = []
    with open(src, "rb") as f:
        while True:
            name = f.readline()
            if not name:
                break
            seq = f.readline().rstrip(b"\n"); f.readline(); qual = f.readline().rstrip(b"\n")
            recs.append((name.rstrip(b"\n"), seq, qual))
    out = []
    for i, (name, seq, qual) in enumerate(recs[:150]):
        plus = b"+"
        if i == 10: plus = b"-"                                   # the '+' line is not one
        if i == 25: qual = qual[: len(qual) // 2]                 # quality shorter than the sequence
        if i == 40: name = name[1:]                               # no '@'
        if i == 55: out.append(b"this line does not belong here")
        if i == 56: out.append(b"")                               # blank line before a name
        if i == 70: qual = qual + b"IIII"                         # quality longer than the sequence
        if i == 85: qual = b"@" + qual[1:]                        # a quality line that starts with '@'
        if i == 100:                                              # a record cut after its sequence line
            out += [name, seq]
            continue
        if i == 120: plus = b"+" + name[1:]                       # '+' followed by the name again: fine
        if i == 130: out += [b"", b""]
        out += [name, seq, plus, qual]
    open(dst, "wb").write(b"\n".join(out) + b"\n")
    out = []
    for i, (name, seq, qual) in enumerate(recs[200:240]):
        plus = b"+"
        if i == 30: plus = b"plus"
        if i == 39:
            out += [name, seq, plus]                              # the last record has no quality line ...
            continue
        out += [name, seq, plus, qual]
    open(dst2, "wb").write(b"\n".join(out))                       # ... and the file no final newline


def main():
    assert os.path.exists(REFBIN), "make -C oracle refbin first"
    os.makedirs(OUT, exist_ok=True)
    ill = os.path.join(HERE, "syn_ill.fq")
    make_illumina_fastq(os.path.join(HERE, "syn.fq"), ill)
    make_malformed_fastq(os.path.join(HERE, "syn.fq"), os.path.join(HERE, "syn_bad.fq"), os.path.join(HERE, "syn_bad2.fq"))
    work = tempfile.mkdtemp()
    for f in os.listdir(HERE):
        if f.startswith("syn."):
            shutil.copy(os.path.join(HERE, f), work)
    shutil.copy(ill, work)
    for f in ("syn_bad.fq", "syn_bad2.fq"):
        shutil.copy(os.path.join(HERE, f), work)
    only = set(sys.argv[1:])                                       # regenerate just these modes (the others keep their committed files)
    shutil.copy(os.path.join(HERE, "subst.txt"), work)
    manifest = json.load(open(os.path.join(OUT, "manifest.json"))) if only else {}
    for name, (args, kw, fq) in MODES.items():
        if only and name not in only:
            continue
        argv = ["-g", "syn.fa", "-o", name, "-a", "0.9"] + args + [fq]
        r = subprocess.run([REFBIN] + argv, cwd=work, capture_output=True, text=True)
        assert r.returncode == 0, (name, r.stderr[-2000:])
        sam = [l for l in open(os.path.join(work, name + ".sam"), "rb") if not l.startswith(b"@PG")]
        with gzip.GzipFile(os.path.join(OUT, name + ".sam.gz"), "wb", mtime=0) as g:
            g.write(b"".join(sam))
        tracks = []
        for ext in ("sgr", "gmp"):
            p = os.path.join(work, name + "." + ext)
            if os.path.exists(p):
                with gzip.GzipFile(os.path.join(OUT, name + "." + ext + ".gz"), "wb", mtime=0) as g:
                    g.write(open(p, "rb").read())
                tracks.append(ext)
        # the same run with 4 threads gives the same SAM set (order is unspecified with -c > 1)
        r4 = subprocess.run([REFBIN, "-g", "syn.fa", "-o", name + "_c4", "-a", "0.9", "-c", "4"] + args + [fq], cwd=work, capture_output=True, text=True)
        assert r4.returncode == 0
        sam4 = [l for l in open(os.path.join(work, name + "_c4.sam"), "rb") if not l.startswith(b"@PG")]
        assert sorted(sam4) == sorted(sam), name
        manifest[name] = dict(argv=args, params=kw, fastq=fq, sam_lines=len(sam), tracks=tracks)
        print(f"{name:26s} {len(sam):5d} SAM lines  {tracks}")
    json.dump(manifest, open(os.path.join(OUT, "manifest.json"), "w"), indent=1, sort_keys=True)
    shutil.rmtree(work)


if __name__ == "__main__":
    main()
