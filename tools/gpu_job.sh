set -e
ROOT=$(pwd); OUT=$ROOT/gpurun_out/fusedpmc; rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp; cd /tmp
for c in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD" \
         "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS"; do
    d="$OUT/sq/$(echo $c | cut -d' ' -f1)"
    rocprofv3 --output-format csv --pmc $c -d "$d" -- python3 "$ROOT/bench.py" --reads 1000000 --steps 2 --cpu-seconds 0 --abi-reads 0 > /dev/null 2> "$OUT/sq.log"
done
for c in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --output-format csv --pmc $c -d "$OUT/pmc/$c" -- python3 "$ROOT/bench.py" --reads 1000000 --steps 2 --cpu-seconds 0 --abi-reads 0 > /dev/null 2> "$OUT/pmc.log"
done
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -- python3 "$ROOT/bench.py" --cpu-seconds 0 --abi-reads 0 > "$OUT/bench_traced.json" 2> "$OUT/stats.log"
cd $ROOT
python3 tools/pmc_sq.py "$OUT/sq" k_ > gpurun_out/fused_sq.txt
python3 tools/summarize_prof.py "$OUT"/stats/*/*_kernel_stats.csv gpurun_out/fused_kernel_stats.csv "$OUT/bench_traced.json"
head -8 gpurun_out/fused_kernel_stats.csv
python3 - <<'PY'
import csv, glob, collections
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    tot = collections.defaultdict(float); cnt = collections.Counter()
    for f in glob.glob(f"gpurun_out/fusedpmc/pmc/{c}/*/*counter_collection.csv"):
        for row in csv.DictReader(open(f)):
            k = row["Kernel_Name"].split("(")[0]
            tot[k] += float(row["Counter_Value"]); cnt[k] += 1
    for k in sorted(tot):
        if any(x in k for x in ("k_vote_tiny", "k_prep", "k_heavy", "k_nw")): print(c, k, "per launch KB-units", tot[k] / cnt[k], "launches", cnt[k])
PY
