# last call of the round: the whole GPU suite at HEAD, then the chrX-shape rows of the bench matrix (k_vote_slots_pp<64> with its table in the filter memory)
set -e
python -m pytest tests -q -m gpu -x > gpurun_out/r4_full_t3.log 2>&1 || { tail -30 gpurun_out/r4_full_t3.log; exit 1; }
tail -2 gpurun_out/r4_full_t3.log
C="--cpu-seconds 5 --abi-reads 4194304"
: > gpurun_out/r04_matrix_part4.jsonl
python3 bench.py --genome-mbp 156 --contigs 1 --mer 10 $C --also="--mer 10 --max-kmer-hits 150 $C" >> gpurun_out/r04_matrix_part4.jsonl 2>> gpurun_out/r04_matrix_part4.log
python3 -c "
import json
for l in open('gpurun_out/r04_matrix_part4.jsonl'):
    j=json.loads(l); print(round(j['value']/1e6,1), j['ms_per_step'], j['kernel_path'][:70], {k:v['ms_per_step'] for k,v in j['kernels'].items()}, j['parity_sample']['mismatches'], j.get('abi_reads_per_s'), j['roofline']['frac'])
"
