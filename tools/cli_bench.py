#!/usr/bin/env python3
"""tools/cli_bench.py — end-to-end rate of the gnumap driver binary (FASTQ file -> SAM + .sgr files) on a synthetic
reference; the numbers quoted in DESIGN.md for SURVEY §8 row (f2).  Usage:
    python tools/cli_bench.py [--mbp 100] [--reads 2000000] [--args "-a 0.9"] [--dir /tmp/gm_cli]
Writes the FASTA / FASTQ with numpy (substitution-only reads, 50 % reverse strand), builds the index through the binary on
its first run, then times a second run (index already on disk)."""
import argparse, os, subprocess, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--mbp", type=float, default=100.0)
    ap.add_argument("--contigs", type=int, default=6)
    ap.add_argument("--reads", type=int, default=2_000_000)
    ap.add_argument("--len", type=int, default=100)
    ap.add_argument("--args", default="-a 0.9")
    ap.add_argument("--dir", default="/tmp/gm_cli")
    ap.add_argument("--sweep", default="", help="further argument sets for the timed run, separated by ';' (each one more run on the same files)")
    a = ap.parse_args()
    os.makedirs(a.dir, exist_ok=True)
    fa = os.path.join(a.dir, "g%g.fa" % a.mbp); fq = os.path.join(a.dir, "r%d.fq" % a.reads)
    bench.make_genome(fa, a.mbp, 42, a.contigs)
    raw = np.fromfile(fa, np.uint8)                                # FASTA text -> 2-bit codes: drop the header lines and the newlines
    keep = np.ones(len(raw), bool)
    for h in np.flatnonzero(raw == ord(">")):
        e = h + int(np.argmax(raw[h:h + 4096] == 10))
        keep[h:e + 1] = False
    keep &= raw != 10
    codes = np.searchsorted(np.frombuffer(b"ACGT", np.uint8), raw[keep]).astype(np.uint8)
    del raw, keep
    rng = np.random.default_rng(7)
    L, n = a.len, a.reads
    acgt = np.frombuffer(b"ACGT", np.uint8)
    with open(fq, "wb") as f:
        for s in range(0, n, 250_000):
            m = min(250_000, n - s)
            pos = rng.integers(0, len(codes) - L - 1, m)
            c = codes[pos[:, None] + np.arange(L)[None, :]].astype(np.int64)
            sub = rng.random((m, L)) < 0.01
            c = np.where(sub, (c + rng.integers(1, 4, (m, L))) % 4, c)
            rev = rng.random(m) < 0.5
            c = np.where(rev[:, None], 3 - c[:, ::-1], c)
            q = (rng.integers(20, 41, (m, L)) + 33).astype(np.uint8)
            rec = np.empty((m, 2 * L + 16), np.uint8)
            names = np.char.zfill((np.arange(s, s + m)).astype(str), 9).astype("S9")
            rec[:, 0] = ord("@"); rec[:, 1] = ord("r"); rec[:, 2:11] = np.frombuffer(names.tobytes(), np.uint8).reshape(m, 9)
            rec[:, 11] = 10; rec[:, 12:12 + L] = acgt[c]; rec[:, 12 + L] = 10; rec[:, 13 + L] = ord("+"); rec[:, 14 + L] = 10
            rec[:, 15 + L:15 + 2 * L] = q; rec[:, 15 + 2 * L] = 10
            f.write(rec.tobytes())
    exe = os.path.join(ROOT, "gnumap_amd", "bin", "gnumap")
    out = os.path.join(a.dir, "out")
    env = dict(os.environ)
    runs = [("index+map", a.args), ("map", a.args)] + [("map " + x.strip(), a.args + " " + x.strip()) for x in a.sweep.split(";") if x.strip()]
    for run, args in runs:
        for ext in (".sam", ".sgr", ".gmp"):                     # a fresh output file each time (overwriting 8 GB of page cache is a different test)
            if os.path.exists(out + ext):
                os.remove(out + ext)
        t0 = time.time()
        r = subprocess.run([exe, "-g", fa, "-o", out, "-v", "1"] + args.split() + [fq], capture_output=True, text=True, env=env)
        dt = time.time() - t0
        print(f"--- {run}: {dt:.2f} s wall, {n / dt / 1e6:.3f} M reads/s end to end (rc {r.returncode})")
        print("\n".join(l for l in r.stderr[-3000:].splitlines() if not l.startswith("[gm_")))
    print("SAM bytes", os.path.getsize(out + ".sam"), "FASTQ bytes", os.path.getsize(fq))


if __name__ == "__main__":
    main()
