set -e
python3 tools/cli_bench.py --mbp 100 --contigs 6 --reads 8000000 --args "-a 0.9" --dir /tmp/gm_cli > gpurun_out/r02_cli_bench_100.txt 2>&1 || { tail -30 gpurun_out/r02_cli_bench_100.txt; exit 1; }
grep -E "^---|wall seconds|stage seconds|Finished" gpurun_out/r02_cli_bench_100.txt
python3 tools/cli_bench.py --mbp 3100 --contigs 24 --reads 8000000 --args "-a 0.9 -m 14" --dir /tmp/gm_cli > gpurun_out/r02_cli_bench_3100.txt 2>&1 || { tail -30 gpurun_out/r02_cli_bench_3100.txt; exit 1; }
grep -E "^---|wall seconds|stage seconds|Finished" gpurun_out/r02_cli_bench_3100.txt
