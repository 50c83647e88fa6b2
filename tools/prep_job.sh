set -e
python -m pytest tests/test_gpu_bucket.py -x -q > gpurun_out/r4_prep_t.log 2>&1 || { tail -30 gpurun_out/r4_prep_t.log; exit 1; }
tail -2 gpurun_out/r4_prep_t.log
C="--cpu-seconds 0 --abi-reads 0"
python bench.py --steps 10 $C > gpurun_out/r4_prep_b.json 2> gpurun_out/r4_prep_b.err
python bench.py --steps 5 --genome-mbp 100 --contigs 6 --mer 12 --jump 6 $C > gpurun_out/r4_prep_c.json 2> gpurun_out/r4_prep_c.err
python -c "
import json
for f in ['r4_prep_b','r4_prep_c']:
  for l in open('gpurun_out/'+f+'.json'):
    j=json.loads(l); print(round(j['value']/1e6,1), j['ms_per_step'], j['kernel_path'][:60], {k:v['ms_per_step'] for k,v in j['kernels'].items()}, j['parity_sample']['mismatches'], j['config'].get('options'))
"
