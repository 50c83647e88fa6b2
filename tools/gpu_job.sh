set -e
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q -k "SEED_FUSED or fused" > gpurun_out/fused_tests9.log 2>&1 || { tail -30 gpurun_out/fused_tests9.log; exit 1; }
tail -2 gpurun_out/fused_tests9.log
GM_SEED_FUSED=1 python3 tools/env_sweep.py --genome-mbp 100 --contigs 6 --mer 12 --sets "" 2>&1 | grep sweep | sed 's/^/[100 m12 fused] /'
GM_SEED_FUSED=1 python3 tools/env_sweep.py --genome-mbp 100 --contigs 6 --mer 14 --sets "" 2>&1 | grep sweep | sed 's/^/[100 m14 fused] /'
python3 tools/env_sweep.py --sets "" 2>&1 | grep sweep | sed 's/^/[3100 m14] /'
