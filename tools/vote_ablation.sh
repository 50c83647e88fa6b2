#!/bin/bash
# tools/vote_ablation.sh <tag>: vector / scalar / LDS instruction counts of k_vote_bucket per read at each early-exit point of the kernel
# (GM_DBG 256: records arrived, 512: headers analysed + routing, 1024: filter pass, 2048: everything but the candidate stores, 0: all)
# -> gpurun_out/<tag>/ablation.txt.  One rocprofv3 --pmc pass per build point (counters only, kernel trace only).
set -eo pipefail
TAG=$1; shift
ROOT=$(pwd); OUT=$ROOT/gpurun_out/$TAG; mkdir -p "$OUT"; export TMPDIR=/tmp; cd /tmp
for dbg in 256 512 1024 2048 0; do
    rocprofv3 --output-format csv --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE -d "$OUT/dbg$dbg" -- python3 "$ROOT/bench.py" --reads 1000000 --steps 2 --cpu-seconds 0 --abi-reads 0 --parity-sample 0 --opt GM_DBG=$dbg "$@" > /dev/null 2>> "$OUT/log.txt" || echo "pass failed: $dbg"
done
cd "$ROOT"
python3 - "$OUT" <<'PY' | tee "$OUT/ablation.txt"
import csv, glob, collections, sys
for dbg in (256, 512, 1024, 2048, 0):
    agg = collections.defaultdict(list)
    for f in glob.glob(f"{sys.argv[1]}/dbg{dbg}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "k_vote_bucket" in r["Kernel_Name"]:
                agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    m = {k: sum(v) / len(v) for k, v in agg.items()}
    print(f"GM_DBG={dbg:5d}  " + "  ".join(f"{k}={m[k] / 1e6:9.2f} per read" if k.startswith("SQ_INSTS") else f"{k}={m[k]:.3e}" for k in sorted(m)))
PY
