set -eo pipefail
ROOT=$(pwd); export TMPDIR=/tmp; cd /tmp
for v in 0 16384 32768; do
  OUT=$ROOT/gpurun_out/pairpmc_$v; rm -rf $OUT; mkdir -p $OUT
  rocprofv3 --output-format csv --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD -d $OUT -- python3 $ROOT/bench.py --reads 1000000 --steps 2 --cpu-seconds 0 --abi-reads 0 --parity-sample 0 --opt GM_DBG=$v > /dev/null 2>> $OUT/log.txt || echo "pass failed $v"
done
cd $ROOT
python3 - <<'PY'
import csv, glob, collections
for v in (0, 16384, 32768):
    agg = collections.defaultdict(list)
    for f in glob.glob(f"gpurun_out/pairpmc_{v}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "k_vote_pair" in r["Kernel_Name"]:
                agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    print("GM_DBG", v, {c: f"{sum(x)/len(x):.4e}" for c, x in sorted(agg.items())})
PY
