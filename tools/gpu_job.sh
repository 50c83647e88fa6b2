set -e
MATRIX=0 bash tools/collect_profiles.sh r02f > gpurun_out/collect_r02f.log 2>&1 || { tail -20 gpurun_out/collect_r02f.log; exit 1; }
tail -2 gpurun_out/collect_r02f.log
: > gpurun_out/matrix_rows_r02f.jsonl
for args in "--genome-mbp 100 --contigs 6 --mer 10" "--genome-mbp 100 --contigs 6 --mer 10 --no-nw" "--genome-mbp 100 --contigs 6 --mer 12" "--genome-mbp 156 --contigs 1 --mer 10" "--genome-mbp 156 --contigs 1 --mer 10 --max-kmer-hits 150"; do
  python3 bench.py $args --cpu-seconds 5 --abi-reads 2097152 >> gpurun_out/matrix_rows_r02f.jsonl 2>> gpurun_out/matrix_rows_r02f.log
done
python3 -c "
import json
for l in open('gpurun_out/matrix_rows_r02f.jsonl'):
    j=json.loads(l); print(round(j['value']/1e6,1), j['ms_per_step'], j['seed_lookup'])"
