set -e
python -m pytest tests -m gpu -x -q > gpurun_out/r02_j17_tests.log 2>&1 || { tail -40 gpurun_out/r02_j17_tests.log; exit 1; }
tail -2 gpurun_out/r02_j17_tests.log
python3 tools/cli_bench.py --mbp 100 --contigs 6 --reads 32000000 --args "-a 0.9" --dir /tmp/gm_cli > gpurun_out/r02_cli_bench_100.txt 2>&1 || { tail -30 gpurun_out/r02_cli_bench_100.txt; exit 1; }
grep -E "^---|wall seconds|stage seconds|Finished|SAM bytes" gpurun_out/r02_cli_bench_100.txt
