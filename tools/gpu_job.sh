set -e
python -m pytest tests/test_gpu_parity.py tests/test_gpu_driver_golden.py tests/test_gpu_scale.py -x -q -k "not every_kernel_variant" > gpurun_out/r02_j15_tests.log 2>&1 || { tail -40 gpurun_out/r02_j15_tests.log; exit 1; }
tail -2 gpurun_out/r02_j15_tests.log
export GM_TRACE=1
timeout -k 10 700 python3 tools/scale_check.py --mbp 1000 --contigs 8 --repeats --mer 14 --reads 200000 --batch 20000 --sample 48 --steps 1 --workdir /tmp/gm_scale_rep > gpurun_out/r02_scale_1000r_nocap.json 2> gpurun_out/r02_scale_1000r_nocap.log || { grep -v "heavy chunk" gpurun_out/r02_scale_1000r_nocap.log | tail -30; exit 1; }
cat gpurun_out/r02_scale_1000r_nocap.json; grep "grouping done\|scale\]" gpurun_out/r02_scale_1000r_nocap.log | tail -14
