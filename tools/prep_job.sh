set -e
python -m pytest tests/test_gpu_parity.py tests/test_gpu_bucket.py tests/test_gpu_celgen.py -x -q -k "not t5 and not t6 and not t7 and not t8 and not t9" > gpurun_out/r4_prep_t.log 2>&1 || { tail -30 gpurun_out/r4_prep_t.log; exit 1; }
tail -2 gpurun_out/r4_prep_t.log
C="--cpu-seconds 0 --abi-reads 0"
python bench.py --steps 10 $C --also="--no-nw $C" --also="--max-kmer-hits 150 $C" > gpurun_out/r4_prep_b.json 2> gpurun_out/r4_prep_b.err
python bench.py --steps 5 --repeats --max-kmer-hits 150 $C > gpurun_out/r4_prep_c.json 2> gpurun_out/r4_prep_c.err
python -c "
import json
for f in ['r4_prep_b','r4_prep_c']:
  for l in open('gpurun_out/'+f+'.json'):
    j=json.loads(l); print(round(j['value']/1e6,1), j['ms_per_step'], {k:v['ms_per_step'] for k,v in j['kernels'].items()}, j['parity_sample']['mismatches'], j['config'].get('options'))
"
