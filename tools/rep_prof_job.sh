set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/rep_prof
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/rep_prof -o rep -- python3 bench.py --steps 3 --warmup 1 --repeats --max-kmer-hits 150 --cpu-seconds 0 --abi-reads 0 --parity-sample 0 > gpurun_out/rep_prof.json 2> gpurun_out/rep_prof.err
python3 - <<'PY'
import csv
rows=list(csv.DictReader(open('gpurun_out/rep_prof/rep_kernel_stats.csv')))
for r in rows[:22]: print(f"{r['Name'][:70]:70s} {r['Calls']:>5} avg {float(r['AverageNs'])/1e3:10.1f} us")
PY
python3 -c "
import json
j=json.loads(open('gpurun_out/rep_prof.json').read().strip().splitlines()[-1]); print(j['value']/1e6, j['ms_per_step'], j['kernels'], j.get('counters_per_step'))"
