# re-entry check: the whole GPU suite at HEAD with its slowest tests, the default bench line, then k_vote_pair's pairs per workgroup
set -e
python -m pytest tests -q -m gpu -x --durations=15 > gpurun_out/r4_full_t4.log 2>&1 || { tail -30 gpurun_out/r4_full_t4.log; exit 1; }
tail -22 gpurun_out/r4_full_t4.log
python3 bench.py > gpurun_out/r4_default_bench.json 2> gpurun_out/r4_default_bench.err
python3 -c "
import json
j=json.loads(open('gpurun_out/r4_default_bench.json').read().strip().splitlines()[-1]); print(round(j['value']/1e6,1), j['ms_per_step'], j['roofline']['frac'], j.get('parity_reference'), j.get('abi_reads_per_s'))
"
bash tools/pair_chunk_job.sh
