set -e
export TMPDIR=/tmp
R=$(pwd)
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r02_j2_stats -- python3 $R/bench.py --genome-mbp 100 --contigs 6 --mer 10 --reads 2000000 --abi-reads 2097152 --cpu-seconds 0 > $R/gpurun_out/r02_j2_bench.json 2> $R/gpurun_out/r02_j2_bench.log || { tail -30 $R/gpurun_out/r02_j2_bench.log; exit 1; }
cd $R
python3 - <<'PY'
import csv,glob
f=glob.glob("gpurun_out/r02_j2_stats/*/*_kernel_stats.csv")[0]
for r in csv.DictReader(open(f)):
    print(r["Name"][:70], r["Calls"], r["TotalDurationNs"], r["AverageNs"])
PY
python3 -c "
import json;j=json.loads(open('gpurun_out/r02_j2_bench.json').read().strip().splitlines()[-1]);print(j['value'],j['abi'])"
