# one box, several builds of the slots kernels (tools/ubench/lib_*.so: scratch builds): configs[1] and chrX-shape rows each
set -e
C="--cpu-seconds 0 --abi-reads 0"
: > gpurun_out/ab_rows.jsonl
cp gnumap_amd/libgnumap_hip.so /tmp/lib_main.so
for f in /tmp/lib_main.so tools/ubench/lib_*.so; do
  cp $f gnumap_amd/libgnumap_hip.so
  echo "{\"lib\": \"$(basename $f)\"}" >> gpurun_out/ab_rows.jsonl
  timeout -k 10 300 python bench.py --steps 3 --genome-mbp 156 --contigs 1 --mer 10 --jump 5 --max-kmer-hits 150 $C >> gpurun_out/ab_rows.jsonl 2>> gpurun_out/ab_rows.err
  timeout -k 10 300 python bench.py --steps 3 --genome-mbp 156 --contigs 1 --mer 10 --jump 5 $C >> gpurun_out/ab_rows.jsonl 2>> gpurun_out/ab_rows.err
done
cp /tmp/lib_main.so gnumap_amd/libgnumap_hip.so
python -c "
import json
for l in open('gpurun_out/ab_rows.jsonl'):
    j=json.loads(l)
    if 'lib' in j: print(j['lib']); continue
    print('   ', round(j['value']/1e6,1), j['ms_per_step'], j['kernel_path'][41:70], j['kernels']['k_vote']['ms_per_step'], j['parity_sample']['mismatches'])
"
