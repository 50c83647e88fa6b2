# slots kernels: parity first, then configs[1] / chrX-shape bench rows with and without the line-aligned copy of the suffix-array runs (GM_ASA)
set -e
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q -k "ASA or (SEED_FUSED and (block or big))" > gpurun_out/pp_t1.log 2>&1 || { tail -30 gpurun_out/pp_t1.log; exit 1; }
tail -2 gpurun_out/pp_t1.log
timeout -k 10 600 python -m pytest tests/test_gpu_scale.py tests/test_gpu_properties.py -x -q -k "configs1 or configs2 or multi_slot or sample_agrees" > gpurun_out/pp_t2.log 2>&1 || { tail -30 gpurun_out/pp_t2.log; exit 1; }
tail -2 gpurun_out/pp_t2.log
C="--cpu-seconds 0 --abi-reads 0"
: > gpurun_out/pp_rows.jsonl
for e in "" "--opt GM_ASA=0" "--opt GM_SLOTS_PIPE=1"; do
  timeout -k 10 300 python bench.py --steps 5 --genome-mbp 100 --contigs 6 --mer 10 --jump 5 $C $e >> gpurun_out/pp_rows.jsonl 2>> gpurun_out/pp_rows.err
done
for e in "" "--opt GM_ASA=0" "--opt GM_SLOTS_PIPE=0"; do
  timeout -k 10 300 python bench.py --steps 3 --genome-mbp 156 --contigs 1 --mer 10 --jump 5 $C $e >> gpurun_out/pp_rows.jsonl 2>> gpurun_out/pp_rows.err
done
python -c "
import json
for l in open('gpurun_out/pp_rows.jsonl'):
    j=json.loads(l); print(round(j['value']/1e6,1), j['ms_per_step'], j['kernel_path'][:70], {k:v['ms_per_step'] for k,v in j['kernels'].items()}, j['parity_sample']['mismatches'], j['config'].get('options'))
"
