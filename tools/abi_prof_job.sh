set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/abi_prof
rocprofv3 --hip-runtime-trace --kernel-trace --memory-copy-trace --stats --output-format csv -d gpurun_out/abi_prof -o abi -- python3 bench.py --steps 1 --warmup 0 --reads 4194304 --cpu-seconds 0 --parity-sample 0 --abi-reads 4194304 --abi-threads 2 --abi-in-flight 6 > gpurun_out/abi_prof.json 2> gpurun_out/abi_prof.err
