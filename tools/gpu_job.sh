set -e
GM_FORCE_DIST=1 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 1 --reads 2000000 --cpu-seconds 0 --abi-reads 0 > gpurun_out/r02_dist1.json 2> gpurun_out/r02_dist1.log || { tail -30 gpurun_out/r02_dist1.log; exit 1; }
python3 -c "
import json;j=json.loads(open('gpurun_out/r02_dist1.json').read().strip().splitlines()[-1]);print('dist1',j['value'],j['coverage_allreduce'])"
python3 bench.py > gpurun_out/r02b_bench_full_10M.json 2> gpurun_out/r02b_bench_full_10M.log || { tail -30 gpurun_out/r02b_bench_full_10M.log; exit 1; }
python3 -c "
import json;j=json.loads(open('gpurun_out/r02b_bench_full_10M.json').read().strip().splitlines()[-1]);print(j['value'],j['ms_per_step'],j['roofline'],j['cpu_baseline']['value'],j['abi_reads_per_s'],{k:v['ms_per_step'] for k,v in j['kernels'].items()})"
