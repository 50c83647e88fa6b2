set -e
C="--cpu-seconds 0 --abi-reads 0"
python bench.py --steps 10 $C > gpurun_out/r4_prep_b.json 2> gpurun_out/r4_prep_b.err
python -c "
import json
for f in ['r4_prep_b']:
  for l in open('gpurun_out/'+f+'.json'):
    j=json.loads(l); print(round(j['value']/1e6,1), j['ms_per_step'], {k:v['ms_per_step'] for k,v in j['kernels'].items()}, j['parity_sample']['mismatches'], j['config'].get('options'))
"
