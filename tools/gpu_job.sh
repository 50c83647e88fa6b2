set -e
python -m pytest tests/test_gpu_parity.py tests/test_gpu_scale.py -x -q -k "not every_kernel_variant or VOTE" > gpurun_out/r02_j20_tests.log 2>&1 || { tail -40 gpurun_out/r02_j20_tests.log; exit 1; }
tail -2 gpurun_out/r02_j20_tests.log
for fx in 1 0; do
GM_VOTE_FIXED=$fx python3 bench.py --genome-mbp 100 --contigs 6 --mer 10 --cpu-seconds 0 --abi-reads 0 2> gpurun_out/r02_j20_c.log | python3 -c "
import json,sys;j=json.loads(sys.stdin.read().strip().splitlines()[-1]);print('c1 fixed=$fx',j['value'],{k:v['ms_per_step'] for k,v in j['kernels'].items()})"
done
python3 bench.py --genome-mbp 156 --contigs 1 --mer 10 --max-kmer-hits 150 --cpu-seconds 0 --abi-reads 0 2> gpurun_out/r02_j20_x.log | python3 -c "
import json,sys;j=json.loads(sys.stdin.read().strip().splitlines()[-1]);print('chrX h150',j['value'],{k:v['ms_per_step'] for k,v in j['kernels'].items()})"
