set -e
timeout -k 10 700 python -m pytest tests/test_gpu_parity.py -x -q -k "every_kernel_variant and VOTE and not VOTE_SLOTS and not VOTE_KERNEL and not wave" > gpurun_out/slots2_tests.log 2>&1 || { tail -30 gpurun_out/slots2_tests.log; exit 1; }
tail -2 gpurun_out/slots2_tests.log
python3 tools/env_sweep.py --genome-mbp 100 --contigs 6 --mer 10 --sets "" "" 2>&1 | grep sweep | sed 's/^/[100 m10] /'
python3 tools/env_sweep.py --genome-mbp 156 --contigs 1 --mer 10 --sets "" 2>&1 | grep sweep | sed 's/^/[156 m10] /'
