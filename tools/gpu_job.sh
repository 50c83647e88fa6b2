set -e
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q -k "SEED_FUSED or fused" > gpurun_out/fused_tests8.log 2>&1 || { tail -30 gpurun_out/fused_tests8.log; exit 1; }
tail -2 gpurun_out/fused_tests8.log
python3 tools/env_sweep.py --genome-mbp 100 --contigs 6 --mer 10 --sets "" "" 2>&1 | grep sweep | sed 's/^/[100 m10] /'
GM_SEED_FUSED=0 python3 tools/env_sweep.py --genome-mbp 100 --contigs 6 --mer 10 --sets "" 2>&1 | grep sweep | sed 's/^/[100 m10 unfused] /'
python3 tools/env_sweep.py --sets "" 2>&1 | grep sweep | sed 's/^/[3100 m14] /'
