set -e
python -m pytest tests/test_gpu_bucket.py tests/test_gpu_celgen.py -x -q -k "(not t1 and not t5 and not t6 and not t7 and not t8 and not t9 and bucket_kernel_matches) or real_sequence" > gpurun_out/r4_t6.log 2>&1 || { tail -20 gpurun_out/r4_t6.log; exit 1; }
tail -2 gpurun_out/r4_t6.log
C="--cpu-seconds 0 --abi-reads 0"
python bench.py --steps 5 $C --also="--opt GM_PAIR_CHUNK=16 $C" --also="--opt GM_PAIR_CHUNK=32 $C" --also="--opt GM_PAIR_CHUNK=128 $C" --also="--opt GM_PAIR_CHUNK=256 $C" --also="--opt GM_VOTE_PAIR=0 $C" > gpurun_out/r4_b5.json 2> gpurun_out/r4_b5.err
python -c "
import json
for l in open('gpurun_out/r4_b5.json'):
    j=json.loads(l); print(round(j['value']/1e6,1), j['ms_per_step'], {k:v['ms_per_step'] for k,v in j['kernels'].items()}, j['parity_sample']['mismatches'], j['config'].get('options'))
"
