set -e
C="--cpu-seconds 0 --abi-reads 0 --steps 3"
python bench.py --genome-mbp 100 --contigs 6 --mer 10 $C --also="--mer 10 --opt GM_NW=lane $C" > gpurun_out/r4_rows_a.jsonl 2> gpurun_out/r4_rows_a.err
python bench.py --genome-mbp 156 --contigs 1 --mer 10 --max-kmer-hits 150 $C > gpurun_out/r4_rows_b.jsonl 2> gpurun_out/r4_rows_b.err
python bench.py --mer 20 --jump 10 --max-kmer-hits 150 $C --also="--read-len 150 --reads 4000000 $C" > gpurun_out/r4_rows_c.jsonl 2> gpurun_out/r4_rows_c.err
python bench.py --repeats --max-kmer-hits 150 $C --also="--mer 20 --jump 10 --max-kmer-hits 150 $C" > gpurun_out/r4_rows_d.jsonl 2> gpurun_out/r4_rows_d.err
cat gpurun_out/r4_rows_?.jsonl | python -c "
import json,sys
for l in sys.stdin:
    j=json.loads(l); print(round(j['value']/1e6,1), j['ms_per_step'], j['config']['workload'][:90], j['kernel_path'], {k:v['ms_per_step'] for k,v in j['kernels'].items()}, j['parity_sample']['mismatches'], j['config'].get('options'))
"
