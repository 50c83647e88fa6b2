# SQ counters of configs[1]'s vote kernel, 1 M reads: k_vote_slots_pp and (GM_SLOTS_PIPE=0) k_vote_slots -> gpurun_out/slots_pmc_{pp,wg}/sq.txt
set -eo pipefail
ROOT=$(pwd); export TMPDIR=/tmp
for v in ${VARIANTS:-pp wg}; do
  OUT=$ROOT/gpurun_out/slots_pmc_$v; mkdir -p "$OUT"; cd /tmp
  E=""; [ $v = wg ] && E="--opt GM_SLOTS_PIPE=0"
  ARGS="--reads 1000000 --steps 2 --cpu-seconds 0 --abi-reads 0 --parity-sample 0 --genome-mbp 100 --contigs 6 --mer 10 --jump 5 $E"
  for c in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD" \
           "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS" \
           "SQ_LDS_ADDR_CONFLICT SQ_LDS_ATOMIC_RETURN SQ_LDS_MEM_VIOLATIONS SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU GRBM_GUI_ACTIVE SQ_INSTS_VMEM_WR"; do
    d="$OUT/sq/$(echo $c | cut -d' ' -f1)"
    timeout -k 10 200 rocprofv3 --output-format csv --pmc $c -d "$d" -- python3 "$ROOT/bench.py" $ARGS > /dev/null 2>> "$OUT/sq.log" || echo "[pmc] pass failed: $c"
  done
  cd "$ROOT"
  python3 tools/pmc_sq.py "$OUT/sq" k_vote > "$OUT/sq.txt"
  cat "$OUT/sq.txt"
done
