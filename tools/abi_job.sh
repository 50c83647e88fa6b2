B="python bench.py --steps 2 --cpu-seconds 0 --parity-sample 0 --reads 12582912 --abi-reads 12582912 --abi-passes 8 --abi-threads 2"
timeout 250 $B --abi-in-flight 3 --abi-block 1048576 > gpurun_out/abi_a.json 2> gpurun_out/abi_a.err
timeout 250 $B --abi-in-flight 2 --abi-block 2097152 > gpurun_out/abi_b.json 2> gpurun_out/abi_b.err
timeout 250 $B --abi-in-flight 2 --abi-block 4194304 > gpurun_out/abi_c.json 2> gpurun_out/abi_c.err
timeout 250 $B --abi-in-flight 3 --abi-block 2097152 > gpurun_out/abi_d.json 2> gpurun_out/abi_d.err
timeout 250 $B --abi-in-flight 4 --abi-block 1048576 > gpurun_out/abi_e.json 2> gpurun_out/abi_e.err
python -c "
import json
for f in ['abi_a','abi_b','abi_c','abi_d','abi_e']:
  try:
   for l in open('gpurun_out/'+f+'.json'):
    j=json.loads(l); a=j.get('abi') or {}; print(f, round(j['value']/1e6,1), a.get('reads_per_s') and round(a['reads_per_s']/1e6,1), a.get('host_threads'), a.get('blocks_in_flight_per_thread'), a.get('block'), a.get('seconds'))
  except Exception as e: print(f, 'failed', e)
"
python bench.py --steps 2 --warmup 1 --genome-mbp 100 --contigs 6 --mer 10 --jump 5 --cpu-seconds 0 --abi-reads 0 --parity-sample 0 --opt GM_DBG=64 > gpurun_out/slots_clk.json 2> gpurun_out/slots_clk.err
grep -i "phase\|tick\|clock" gpurun_out/slots_clk.err | tail -12
