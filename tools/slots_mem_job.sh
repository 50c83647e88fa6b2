# HBM traffic of configs[1]'s vote kernel (1 M reads per launch): FETCH_SIZE / WRITE_SIZE / L2 hits / EA read requests, one counter set per run
set -eo pipefail
ROOT=$(pwd); export TMPDIR=/tmp; OUT=$ROOT/gpurun_out/slots_mem; rm -rf "$OUT"; mkdir -p "$OUT"; cd /tmp
ARGS="--reads 1000000 --steps 2 --cpu-seconds 0 --abi-reads 0 --parity-sample 0 --genome-mbp 100 --contigs 6 --mer 10 --jump 5"
for c in FETCH_SIZE WRITE_SIZE TCC_EA0_RDREQ_sum "TCC_HIT_sum TCC_MISS_sum" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum"; do
    d="$OUT/mem/$(echo $c | tr ' ' '_')"
    timeout -k 10 200 rocprofv3 --output-format csv --pmc $c -d "$d" -- python3 "$ROOT/bench.py" $ARGS > /dev/null 2>> "$OUT/mem.log" || echo "[pmc] pass failed: $c"
done
cd "$ROOT"
python3 tools/pmc_sq.py "$OUT/mem" k_vote > "$OUT/mem.txt"
cat "$OUT/mem.txt"
