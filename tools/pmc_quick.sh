#!/bin/bash
# tools/pmc_quick.sh <tag> [bench args...]: SQ + memory counter passes of bench.py at 1 M reads (each counter set in its own run, kernel
# trace only) -> gpurun_out/<tag>/{sq.txt,mem.txt}
set -eo pipefail
TAG=$1; shift
ROOT=$(pwd); OUT=$ROOT/gpurun_out/$TAG; mkdir -p "$OUT"; export TMPDIR=/tmp; cd /tmp
ARGS="--reads 1000000 --steps 2 --cpu-seconds 0 --abi-reads 0 --parity-sample 0 $*"
for c in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD" \
         "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS" \
         "SQ_INSTS_BRANCH SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU GRBM_GUI_ACTIVE"; do
    d="$OUT/sq/$(echo $c | cut -d' ' -f1)"
    rocprofv3 --output-format csv --pmc $c -d "$d" -- python3 "$ROOT/bench.py" $ARGS > /dev/null 2>> "$OUT/sq.log" || echo "[pmc] pass failed: $c"
done
for c in FETCH_SIZE WRITE_SIZE TCC_EA0_RDREQ_sum "TCC_HIT_sum TCC_MISS_sum"; do
    d="$OUT/mem/$(echo $c | tr ' ' '_')"
    rocprofv3 --output-format csv --pmc $c -d "$d" -- python3 "$ROOT/bench.py" $ARGS > /dev/null 2>> "$OUT/mem.log" || echo "[pmc] pass failed: $c"
done
cd "$ROOT"
python3 tools/pmc_sq.py "$OUT/sq" k_ > "$OUT/sq.txt"
python3 tools/pmc_sq.py "$OUT/mem" k_ > "$OUT/mem.txt"
grep -A 30 "k_vote" "$OUT/sq.txt" | head -60
