set -e
python -m pytest tests/test_gpu_parity.py tests/test_gpu_driver_golden.py -x -q -k "not every_kernel_variant or HEAVY" > gpurun_out/r02_j10_tests.log 2>&1 || { tail -40 gpurun_out/r02_j10_tests.log; exit 1; }
tail -2 gpurun_out/r02_j10_tests.log
python3 bench.py --cpu-seconds 0 --abi-reads 0 2> gpurun_out/r02_j10_h.log | python3 -c "
import json,sys;j=json.loads(sys.stdin.read().strip().splitlines()[-1]);print('human',j['value'],{k:v['ms_per_step'] for k,v in j['kernels'].items()})"
timeout -k 10 500 python3 tools/scale_check.py --mbp 1000 --contigs 8 --repeats --mer 14 --reads 200000 --sample 48 --steps 2 --workdir /tmp/gm_scale_rep > gpurun_out/r02_scale_1000r_nocap.json 2> gpurun_out/r02_scale_1000r_nocap.log || { tail -20 gpurun_out/r02_scale_1000r_nocap.log; exit 1; }
cat gpurun_out/r02_scale_1000r_nocap.json
