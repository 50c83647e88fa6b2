# quick A/B of the slots kernels: configs[1] and the chrX shape, with / without the line-aligned copy of the suffix-array runs; GM_TRACE shows what was built
set -e
C="--cpu-seconds 0 --abi-reads 0"
: > gpurun_out/pp_rows.jsonl; : > gpurun_out/pp_rows.err
for e in "" "--opt GM_ASA=0"; do
  GM_TRACE=1 timeout -k 10 300 python bench.py --steps 5 --genome-mbp 100 --contigs 6 --mer 10 --jump 5 $C $e >> gpurun_out/pp_rows.jsonl 2>> gpurun_out/pp_rows.err
done
for e in "" "--opt GM_ASA=0"; do
  timeout -k 10 300 python bench.py --steps 3 --genome-mbp 156 --contigs 1 --mer 10 --jump 5 $C $e >> gpurun_out/pp_rows.jsonl 2>> gpurun_out/pp_rows.err
done
grep -i "aligned suffix\|path:" gpurun_out/pp_rows.err | sort | uniq -c | head -8
python -c "
import json
for l in open('gpurun_out/pp_rows.jsonl'):
    j=json.loads(l); print(round(j['value']/1e6,1), j['ms_per_step'], j['kernel_path'][:70], {k:v['ms_per_step'] for k,v in j['kernels'].items()}, j['parity_sample']['mismatches'], j['config'].get('options'))
"
