set -e
export GM_TRACE=1
timeout -k 10 400 python3 tools/scale_check.py --mbp 1000 --contigs 8 --repeats --mer 14 --reads 20000 --sample 8 --steps 1 --keep --workdir /tmp/gm_scale_rep > gpurun_out/r02_scale_1000r_20k.json 2> gpurun_out/r02_scale_1000r_20k.log || { grep -v "heavy chunk" gpurun_out/r02_scale_1000r_20k.log | tail -30; exit 1; }
cat gpurun_out/r02_scale_1000r_20k.json; grep -v "heavy chunk" gpurun_out/r02_scale_1000r_20k.log | tail -30
