set -e
echo skip tests

python tools/cli_bench.py --mbp 3100 --contigs 24 --reads 32000000 --args "-a 0.9 -m 14 -j 7" --sweep="--sam_write=pwrite;--workers=4;--fmt_threads=6;--fmt_threads=12" > gpurun_out/r04_cli_bench_3100Mbp_32M.txt 2>&1
cat gpurun_out/r04_cli_bench_3100Mbp_32M.txt | grep -v "^gnumap-mi355x\|^Finished" | cut -c1-220
