set -e
for w in 3 5 8; do
python3 tools/cli_bench.py --mbp 100 --contigs 6 --reads 32000000 --args "-a 0.9 --workers=$w" --dir /tmp/gm_cli > gpurun_out/r02_cli_w$w.txt 2>&1 || { tail -30 gpurun_out/r02_cli_w$w.txt; exit 1; }
echo "workers=$w"; grep -E "wall seconds|stage seconds" gpurun_out/r02_cli_w$w.txt
done
