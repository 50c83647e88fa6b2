#!/bin/bash
# tools/cli_sweep.sh — FASTQ->SAM pipeline rate of the driver binary for a few --workers / --batch settings (uses the files
# tools/cli_bench.py left in /tmp/gm_cli)
for w in 2 3 4; do for bsz in 262144 524288; do
  ./gnumap_amd/bin/gnumap -g /tmp/gm_cli/g100.fa -o /tmp/gm_cli/out -a 0.9 -v 1 --workers=$w --batch=$bsz /tmp/gm_cli/r8000000.fq 2>&1 | grep "wall seconds" | sed "s/^/workers=$w batch=$bsz: /"
done; done
