#!/bin/bash
# Collect the measurements kept under profiles/ (run on an MI355X box from the repo root):
#   tools/collect_profiles.sh <tag>        e.g. r01b  ->  profiles/<tag>_*  and profiles/pmc_latest.json
# Kernel trace + stats, the PMC passes (each counter set in its own run, never mixed with a trace domain other than the kernel
# trace), the sampled phase clocks of the vote kernel, and a plain bench.py line.  Raw output goes to gpurun_out/<tag>/.
set -eo pipefail
TAG=${1:-rNN}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/$TAG
rm -rf "$OUT"; mkdir -p "$OUT"
export TMPDIR=/tmp
KEY=${KEY:-g3100m_c24_s42_L100_m14_full}    # workload key of profiles/pmc_latest.json (bench.py matches it): the default = human-scale workload
# configs[1]: BENCH_ARGS="--genome-mbp 100 --contigs 6 --mer 10" KEY=g100m_c6_s42_L100_m10_full MATRIX=0 LATEST=0 tools/collect_profiles.sh r03_c1
cd /tmp                                            # rocprofv3 scratch files go to the cwd
B="python3 $ROOT/bench.py ${BENCH_ARGS:-}"
if [ "${MATRIX:-1}" != "only" ]; then
echo "[collect] plain bench"
$B > "$OUT/bench.json" 2> "$OUT/bench.log"
echo "[collect] kernel trace + stats"
# the same workload and timed region as the plain run; only the two side legs (CPU baseline, ABI leg: extra launches of the same
# kernels on 262 144-read blocks) are off so that the per-kernel averages are those of the timed steps
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -- python3 "$ROOT/bench.py" ${BENCH_ARGS:-} --cpu-seconds 0 --abi-reads 0 --parity-sample 0 > "$OUT/bench_traced.json" 2> "$OUT/stats.log"
for c in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum" TCC_EA0_RDREQ_sum; do
    d="$OUT/pmc/$(echo $c | tr ' ' '_')"
    echo "[collect] pmc $c"
    rocprofv3 --output-format csv --pmc $c -d "$d" -- python3 "$ROOT/bench.py" ${BENCH_ARGS:-} --reads 1000000 --steps 2 --cpu-seconds 0 --abi-reads 0 --parity-sample 0 > /dev/null 2> "$OUT/pmc.log"
done
for c in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD" \
         "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS"; do
    d="$OUT/sq/$(echo $c | cut -d' ' -f1)"
    echo "[collect] pmc $c"
    rocprofv3 --output-format csv --pmc $c -d "$d" -- python3 "$ROOT/bench.py" ${BENCH_ARGS:-} --reads 1000000 --steps 2 --cpu-seconds 0 --abi-reads 0 --parity-sample 0 > /dev/null 2> "$OUT/sq.log" || echo "[collect] SQ pass failed (see sq.log)"
done
echo "[collect] vote phase clocks"
GM_DBG=64 $B --genome-mbp 100 --contigs 6 --mer 10 --reads 1000000 --cpu-seconds 0 --abi-reads 0 > /dev/null 2> "$OUT/phase.log"
fi
if [ "${MATRIX:-1}" = "1" ] || [ "${MATRIX:-1}" = "only" ]; then
    echo "[collect] bench matrix over the BASELINE configurations (10 M reads per step, CPU baseline 5 s each; rows that share a reference and its reads run in one process: --also)"
    : > "$OUT/matrix.jsonl"
    C="--cpu-seconds 5 --abi-reads 4194304"
    run() { echo "[collect]   bench.py $*"; $B "$@" >> "$OUT/matrix.jsonl" 2>> "$OUT/matrix.log"; }
    if [ "${MATRIX_PART:-all}" != "2" ]; then
    # human scale, i.i.d.: the default run, --no_nw (configs[4] shape), the flags SURVEY 8(d) suggests for human
    run $C --also="--no-nw $C" --also="--mer 20 --jump 10 --max-kmer-hits 150 $C" --also="--mer 20 --jump 10 --max-kmer-hits 150 --no-nw $C" --also="--max-kmer-hits 150 $C" --also="--opt GM_VOTE_PAIR=0 $C" --also="--opt GM_NW=lane $C"
    # human scale, repeat-rich: capped runs, NW and --no_nw
    run --repeats --max-kmer-hits 150 $C --also="--max-kmer-hits 150 --no-nw $C" --also="--mer 20 --jump 10 --max-kmer-hits 150 $C" --also="--mer 20 --jump 10 --max-kmer-hits 150 --no-nw $C"
    fi
    if [ "${MATRIX_PART:-all}" != "1" ]; then
    run --read-len 150 --reads 4000000 $C                                        # configs[3] read length
    run --genome-mbp 100 --contigs 6 --mer 10 $C --also="--mer 10 --no-nw $C" --also="--mer 12 $C" --also="--mer 14 $C" --also="--mer 16 --jump 8 $C" --also="--mer 20 --jump 10 $C"
    run --genome-mbp 156 --contigs 1 --mer 10 $C --also="--mer 10 --max-kmer-hits 150 $C" --also="--mer 16 --jump 8 $C"
    fi
fi
cd "$ROOT"
[ -s "$OUT/matrix.jsonl" ] && cp "$OUT/matrix.jsonl" profiles/${TAG}_bench_matrix_10M${MATRIX_PART:+_part$MATRIX_PART}.jsonl
if [ "${MATRIX:-1}" = "only" ]; then echo "[collect] matrix done"; exit 0; fi
python3 tools/summarize_prof.py "$OUT"/stats/*/*_kernel_stats.csv profiles/${TAG}_kernel_stats_10M.csv "$OUT/bench_traced.json"
if [ "${LATEST:-1}" = "1" ]; then PJ=profiles/pmc_latest.json; else PJ="$OUT/pmc_other.json"; fi
python3 tools/pmc_summary.py "$OUT/pmc" 1000000 $KEY profiles/${TAG}_pmc_1M.txt $PJ
python3 tools/pmc_sq.py "$OUT/sq" k_ > profiles/${TAG}_sq_counters_1M.txt
grep "gm_dbg" "$OUT/phase.log" > profiles/${TAG}_vote_phase_clocks_1M.txt || true
tail -1 "$OUT/bench.json" > profiles/${TAG}_bench_full_10M.json
cp profiles/${TAG}_*  profiles/pmc_latest.json "$OUT/" 2>/dev/null || true
echo "[collect] done"
