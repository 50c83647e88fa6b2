set -e
python -m pytest tests/test_gpu_driver_golden.py tests/test_gpu_parity.py tests/test_gpu_scale.py -x -q -k "not every_kernel_variant" > gpurun_out/r02_j5_tests.log 2>&1 || { tail -40 gpurun_out/r02_j5_tests.log; exit 1; }
tail -3 gpurun_out/r02_j5_tests.log
GM_TIMING=1 python3 bench.py --cpu-seconds 0 > gpurun_out/r02_j5_bench_human.json 2> gpurun_out/r02_j5_bench_human.log || { tail -30 gpurun_out/r02_j5_bench_human.log; exit 1; }
python3 -c "
import json;j=json.loads(open('gpurun_out/r02_j5_bench_human.json').read().strip().splitlines()[-1]);print(j['value'],j['abi'])"
grep -E "gm_timing" gpurun_out/r02_j5_bench_human.log | tail -6
