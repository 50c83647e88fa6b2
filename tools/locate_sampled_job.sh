#!/bin/bash
# tools/locate_sampled_job.sh <tag>: the path north_star names - FM-index backward search (k_seed) + sampled-SA locate by LF walks
# (k_locate_sampled, src/bwt.c:86-96) - measured at 3.1 Gbp: bench lines (roofline of the dominant kernel from the kernel-side work
# counters), rocprofv3 kernel stats and the HBM counters of the same commands -> gpurun_out/<tag>/ (copy what is to be kept into profiles/)
set -eo pipefail
TAG=$1
ROOT=$(pwd); OUT=$ROOT/gpurun_out/$TAG; mkdir -p "$OUT"; export TMPDIR=/tmp; cd /tmp
A14="--locate sampled --reads 1000000 --steps 3 --abi-reads 0 --cpu-seconds 5"
A20="--locate sampled --reads 4000000 --steps 3 --abi-reads 0 --cpu-seconds 5 --mer 20 --jump 10 --max-kmer-hits 150"
python3 "$ROOT/bench.py" $A14 > "$OUT/bench_m14.json" 2> "$OUT/bench_m14.log"
python3 "$ROOT/bench.py" $A20 > "$OUT/bench_m20.json" 2> "$OUT/bench_m20.log"
for cfg in m14 m20; do
    if [ $cfg = m14 ]; then A="$A14"; else A="$A20"; fi
    rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats_$cfg" -- python3 "$ROOT/bench.py" $A --cpu-seconds 0 --parity-sample 0 > "$OUT/traced_$cfg.json" 2> "$OUT/stats_$cfg.log"
    for c in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum" TCC_EA0_RDREQ_sum; do
        d="$OUT/pmc_$cfg/$(echo $c | tr ' ' '_')"
        rocprofv3 --output-format csv --pmc $c -d "$d" -- python3 "$ROOT/bench.py" $A --steps 2 --cpu-seconds 0 --parity-sample 0 > /dev/null 2>> "$OUT/pmc_$cfg.log"
    done
done
cd "$ROOT"
for cfg in m14 m20; do
    python3 tools/summarize_prof.py "$OUT"/stats_$cfg/*/*_kernel_stats.csv "$OUT/kernel_stats_$cfg.csv" "$OUT/traced_$cfg.json" || true
    python3 tools/pmc_sq.py "$OUT/pmc_$cfg" k_ > "$OUT/pmc_$cfg.txt" || true
done
cat "$OUT/bench_m14.json" "$OUT/bench_m20.json" | python3 -c "
import json,sys
for l in sys.stdin:
    j=json.loads(l); print(round(j['value']/1e6,2),'M reads/s', j['ms_per_step'],'ms', j['kernel_path'], j['roofline'], {k:v['ms_per_step'] for k,v in j['kernels'].items()}, j['parity_sample'])
"
