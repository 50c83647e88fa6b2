set -e
export GM_TRACE=1
timeout -k 10 300 python3 tools/scale_check.py --mbp 1000 --contigs 8 --repeats --mer 14 --reads 20000 --sample 4 --steps 1 --workdir /tmp/gm_scale_rep > gpurun_out/r02_j13.json 2> gpurun_out/r02_j13.log || { grep -v "heavy chunk" gpurun_out/r02_j13.log | tail -30; exit 1; }
grep "grouping done" gpurun_out/r02_j13.log
