set -e
python -m pytest tests/test_gpu_driver_golden.py tests/test_gpu_parity.py -x -q -k "not every_kernel_variant" > gpurun_out/r02_j19_tests.log 2>&1 || { tail -40 gpurun_out/r02_j19_tests.log; exit 1; }
tail -2 gpurun_out/r02_j19_tests.log
for i in 1 2; do
GM_TIMING=1 python3 bench.py --cpu-seconds 0 2> gpurun_out/r02_j19_h.log | python3 -c "
import json,sys;j=json.loads(sys.stdin.read().strip().splitlines()[-1]);print('human',j['value'],j['abi'])"
done
grep gm_timing gpurun_out/r02_j19_h.log | tail -4
