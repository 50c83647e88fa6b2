set -e
python -m pytest tests/test_gpu_nw_rows.py tests/test_gpu_celgen.py tests/test_gpu_driver_golden.py -x -q > gpurun_out/r4_prep_t.log 2>&1 || { tail -30 gpurun_out/r4_prep_t.log; exit 1; }
tail -2 gpurun_out/r4_prep_t.log
C="--cpu-seconds 0 --abi-reads 0"
python bench.py --steps 5 $C --also="--opt GM_PREP=tile $C" --also="--read-len 150 --reads 4000000 $C" --also="--read-len 150 --reads 4000000 --opt GM_PREP=tile $C" --also="--opt GM_VOTE_PAIR=0 $C" > gpurun_out/r4_prep_b.json 2> gpurun_out/r4_prep_b.err
python -c "
import json
for l in open('gpurun_out/r4_prep_b.json'):
    j=json.loads(l); print(round(j['value']/1e6,1), j['ms_per_step'], {k:v['ms_per_step'] for k,v in j['kernels'].items()}, j['parity_sample']['mismatches'], j['config'].get('options'))
"
GM_TRACE=1 python bench.py --steps 1 --warmup 0 --reads 1000000 --cpu-seconds 0 --parity-sample 0 --abi-reads 524288 --abi-threads 1 --abi-in-flight 1 > gpurun_out/abi_trace.json 2> gpurun_out/abi_trace.err || true
grep -c gm_trace gpurun_out/abi_trace.err
