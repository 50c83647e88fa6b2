python -m pytest tests/test_gpu_parity.py tests/test_gpu_driver_golden.py tests/test_gpu_abi_async.py -x -q -k "traceback or cigar or sam or cli or driver or abi or output" > gpurun_out/r4_tb_t.log 2>&1 || { tail -30 gpurun_out/r4_tb_t.log; exit 1; }
tail -2 gpurun_out/r4_tb_t.log
B="python bench.py --steps 2 --cpu-seconds 0 --parity-sample 0 --abi-reads 8388608 --abi-passes 12 --abi-threads 2 --abi-in-flight 6"
export GPU_MAX_HW_QUEUES=16
timeout 200 $B > gpurun_out/abi_f.json 2> gpurun_out/abi_f.err
timeout 200 $B --opt GM_TRACEBACK=direct > gpurun_out/abi_g.json 2> gpurun_out/abi_g.err
GPU_FORCE_BLIT_COPY_SIZE=0 timeout 200 $B > gpurun_out/abi_a.json 2> gpurun_out/abi_a.err
DEBUG_CLR_LIMIT_BLIT_WG=64 timeout 200 $B > gpurun_out/abi_c.json 2> gpurun_out/abi_c.err
HSA_ENABLE_SDMA=1 GPU_BLIT_ENGINE_TYPE=2 timeout 200 $B > gpurun_out/abi_d.json 2> gpurun_out/abi_d.err
DEBUG_HIP_DYNAMIC_QUEUES=1 timeout 200 $B > gpurun_out/abi_e.json 2> gpurun_out/abi_e.err
python -c "
import json
for f in ['abi_f','abi_g','abi_a','abi_c','abi_d','abi_e']:
  try:
   for l in open('gpurun_out/'+f+'.json'):
    j=json.loads(l); a=j.get('abi') or {}; print(f, round(j['value']/1e6,1), j['config'].get('options'), a.get('reads_per_s') and round(a['reads_per_s']/1e6,1), a.get('host_threads'), a.get('blocks_in_flight_per_thread'), a.get('block'), a.get('seconds'))
  except Exception as e: print(f, 'failed', e)
"
