set -e
python -m pytest tests/test_gpu_parity.py tests/test_gpu_scale.py -x -q -k "not every_kernel_variant or VOTE_SLOTS" > gpurun_out/r02_j21_tests.log 2>&1 || { tail -40 gpurun_out/r02_j21_tests.log; exit 1; }
tail -2 gpurun_out/r02_j21_tests.log
python3 bench.py --cpu-seconds 0 --abi-reads 0 2> gpurun_out/r02_j21_h.log | python3 -c "
import json,sys;j=json.loads(sys.stdin.read().strip().splitlines()[-1]);print('human',j['value'],{k:v['ms_per_step'] for k,v in j['kernels'].items()})"
python3 bench.py --genome-mbp 100 --contigs 6 --mer 12 --cpu-seconds 0 --abi-reads 0 2> gpurun_out/r02_j21_m12.log | python3 -c "
import json,sys;j=json.loads(sys.stdin.read().strip().splitlines()[-1]);print('100Mbp m12',j['value'],{k:v['ms_per_step'] for k,v in j['kernels'].items()})"
