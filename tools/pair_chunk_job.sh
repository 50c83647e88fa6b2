# k_vote_pair: pairs per workgroup (GM_PAIR_CHUNK), one box, the default run
set -e
C="--cpu-seconds 0 --abi-reads 0"
: > gpurun_out/pc_rows.jsonl
timeout -k 10 400 python bench.py --steps 20 $C --also="--opt GM_PAIR_CHUNK=16 $C" --also="--opt GM_PAIR_CHUNK=64 $C" --also="--opt GM_PAIR_CHUNK=128 $C" --also="--opt GM_PAIR_CHUNK=32 $C" >> gpurun_out/pc_rows.jsonl 2>> gpurun_out/pc_rows.err
python -c "
import json
for l in open('gpurun_out/pc_rows.jsonl'):
    j=json.loads(l); print(round(j['value']/1e6,1), j['ms_per_step'], {k:v['ms_per_step'] for k,v in j['kernels'].items()}, j['parity_sample']['mismatches'], j['config'].get('options'))
"
