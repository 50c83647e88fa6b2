set -e
C="--cpu-seconds 0 --abi-reads 0"
python bench.py --steps 5 --read-len 150 --reads 4000000 $C --also="--opt GM_PREP=tile $C" > gpurun_out/r4_l150.json 2> gpurun_out/r4_l150.err
python bench.py --steps 3 --cpu-seconds 0 --parity-sample 0 --abi-reads 8388608 --abi-threads 2 --abi-in-flight 6 > gpurun_out/abi_a.json 2> gpurun_out/abi_a.err
python bench.py --steps 3 --cpu-seconds 0 --parity-sample 0 --abi-reads 8388608 --abi-threads 4 --abi-in-flight 3 > gpurun_out/abi_b.json 2> gpurun_out/abi_b.err
python bench.py --steps 3 --cpu-seconds 0 --parity-sample 0 --abi-reads 8388608 --abi-threads 2 --abi-in-flight 3 --abi-block 1048576 > gpurun_out/abi_c.json 2> gpurun_out/abi_c.err
python bench.py --steps 3 --cpu-seconds 0 --parity-sample 0 --abi-reads 8388608 --abi-threads 1 --abi-in-flight 8 > gpurun_out/abi_d.json 2> gpurun_out/abi_d.err
python bench.py --steps 3 --cpu-seconds 0 --parity-sample 0 --abi-reads 8388608 --abi-threads 8 --abi-in-flight 2 > gpurun_out/abi_e.json 2> gpurun_out/abi_e.err
python -c "
import json
for f in ['r4_l150','abi_a','abi_b','abi_c','abi_d','abi_e']:
  for l in open('gpurun_out/'+f+'.json'):
    j=json.loads(l); a=j.get('abi') or {}; print(f, round(j['value']/1e6,1), j['ms_per_step'], {k:v['ms_per_step'] for k,v in j['kernels'].items()}, j['config'].get('options'), a.get('reads_per_s') and round(a['reads_per_s']/1e6,1), a.get('host_threads'), a.get('blocks_in_flight_per_thread'), a.get('block'))
"
