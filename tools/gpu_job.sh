set -e
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -k "nw_score_bits or band_width or map_batch_matches or long_reads or illumina or bad_quality" > gpurun_out/nwlds_tests.log 2>&1 || { tail -30 gpurun_out/nwlds_tests.log; exit 1; }
tail -2 gpurun_out/nwlds_tests.log
python3 tools/env_sweep.py --sets "" "" 2>&1 | grep sweep | sed 's/^/[3100 m14] /'
python3 tools/env_sweep.py --genome-mbp 100 --contigs 6 --mer 10 --sets "" 2>&1 | grep sweep | sed 's/^/[100 m10] /'
