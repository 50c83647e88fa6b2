C="--cpu-seconds 0 --abi-reads 0 --parity-sample 0 --steps 5"
python bench.py $C --opt GM_PAIR_GRID=32768 --also="--opt GM_PAIR_GRID=32768 --opt GM_DBG=4096 $C" --also="--opt GM_PAIR_GRID=32768 --opt GM_DBG=8192 $C" --also="--opt GM_PAIR_GRID=32768 --opt GM_DBG=12288 $C" --also="--opt GM_PAIR_GRID=32768 --opt GM_DBG=16384 $C" --also="--opt GM_PAIR_GRID=32768 --opt GM_DBG=32768 $C" --also="--opt GM_PAIR_GRID=32768 --opt GM_DBG=36864 $C" > gpurun_out/r4_b6.json 2> gpurun_out/r4_b6.err
python -c "
import json
for l in open('gpurun_out/r4_b6.json'):
    j=json.loads(l); print(j['kernels']['k_vote']['ms_per_step'], j['config'].get('options'))
"
