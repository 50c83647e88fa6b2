#!/bin/bash
# tools/pmc_kernel.sh <tag> <kernel substring> [bench args...]: SQ counters of one kernel at 1 M reads per launch -> gpurun_out/<tag>/counters.txt
set -eo pipefail
TAG=$1; KN=$2; shift; shift
ROOT=$(pwd); OUT=$ROOT/gpurun_out/$TAG; mkdir -p "$OUT"; export TMPDIR=/tmp; cd /tmp
ARGS="--reads 1000000 --steps 2 --cpu-seconds 0 --abi-reads 0 --parity-sample 0 $*"
for c in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD" \
         "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS" \
         "GRBM_GUI_ACTIVE TCC_EA0_RDREQ_sum"; do
    d="$OUT/$(echo $c | cut -d' ' -f1)"
    rocprofv3 --output-format csv --pmc $c -d "$d" -- python3 "$ROOT/bench.py" $ARGS > /dev/null 2>> "$OUT/log.txt" || echo "[pmc] pass failed: $c"
done
cd "$ROOT"
python3 - "$OUT" "$KN" <<'PY' | tee "$OUT/counters.txt"
import csv, glob, collections, sys
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(f"{sys.argv[1]}/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if sys.argv[2] in r["Kernel_Name"]:
            agg[r["Kernel_Name"].split("(")[0][-40:]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in agg.items():
    print(k)
    for c, x in sorted(v.items()):
        print(f"    {c:26s} {sum(x) / len(x):14.4e}  ({len(x)} launches)")
PY
