# ABI leg under the HIP runtime's copy-path switches (the leg's uploads run as blit kernels at 34 GB/s: profiles/r04_abi_timeline.txt).
# 100 Mbp / -m 12 (k_vote_pair, same per-block device work as the headline; the leg is copy-bound), one process per setting.
B="python3 bench.py --genome-mbp 100 --contigs 6 --mer 12 --steps 2 --cpu-seconds 0 --parity-sample 16 --reads 8388608 --abi-reads 8388608 --abi-passes 6"
run() {  # name, env assignments...
  name=$1; shift
  env "$@" timeout -k 10 200 $B > gpurun_out/abienv_$name.json 2> gpurun_out/abienv_$name.err || echo "$name: rc $?"
}
run base       GM_NOOP=1
run wg4        DEBUG_CLR_LIMIT_BLIT_WG=4
run wg64       DEBUG_CLR_LIMIT_BLIT_WG=64
run wg256      DEBUG_CLR_LIMIT_BLIT_WG=256
run nosdma     HSA_ENABLE_SDMA=0
run blitall    GPU_FORCE_BLIT_COPY_SIZE=4194304
run blitnone   GPU_FORCE_BLIT_COPY_SIZE=0
run base2      GM_NOOP=1
python3 - <<'PY'
import json, glob
for n in ["base","wg4","wg64","wg256","nosdma","blitall","blitnone","base2"]:
    try:
        j = json.loads(open(f"gpurun_out/abienv_{n}.json").read().strip().splitlines()[-1])
        a = j.get("abi") or {}
        print(n, "value", round(j["value"]/1e6,1), "abi", round((j.get("abi_reads_per_s") or 0)/1e6,1), "pcie_incl", round(j["pcie_inclusive_reads_per_s"]/1e6,1), j["parity_sample"]["mismatches"])
    except Exception as e:
        print(n, "failed", e)
PY
