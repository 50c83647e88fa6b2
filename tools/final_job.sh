# round-end collection: headline profiles at HEAD (collect_profiles.sh, no matrix) + the bench-matrix rows this round's last kernels changed
set -e
MATRIX=0 bash tools/collect_profiles.sh r04 > gpurun_out/collect_r04.log 2>&1 || { tail -20 gpurun_out/collect_r04.log; exit 1; }
tail -3 gpurun_out/collect_r04.log
C="--cpu-seconds 5 --abi-reads 4194304"
: > gpurun_out/r04_matrix_part3.jsonl
python3 bench.py --genome-mbp 100 --contigs 6 --mer 10 $C --also="--mer 10 --no-nw $C" >> gpurun_out/r04_matrix_part3.jsonl 2>> gpurun_out/r04_matrix_part3.log
python3 bench.py --genome-mbp 156 --contigs 1 --mer 10 $C --also="--mer 10 --max-kmer-hits 150 $C" >> gpurun_out/r04_matrix_part3.jsonl 2>> gpurun_out/r04_matrix_part3.log
cp gpurun_out/r04_matrix_part3.jsonl profiles/r04_bench_matrix_10M_part3.jsonl
python3 -c "
import json
for l in open('gpurun_out/r04_matrix_part3.jsonl'):
    j=json.loads(l); print(round(j['value']/1e6,1), j['ms_per_step'], j['kernel_path'][:70], {k:v['ms_per_step'] for k,v in j['kernels'].items()}, j['parity_sample']['mismatches'], j.get('abi_reads_per_s'), j['roofline']['frac'])
"
python3 -c "
import json
j=json.loads(open('profiles/r04_bench_full_10M.json').read()); print('HEADLINE', round(j['value']/1e6,1), j['ms_per_step'], j['roofline'], j.get('parity_reference'), j.get('abi_reads_per_s'))
"
