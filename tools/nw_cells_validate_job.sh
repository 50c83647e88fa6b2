# k_nw_rows with LDS-read cell values: the whole GPU suite, then the default bench line and the 150-bp row (k_nw_rows<19>)
set -e
python -m pytest tests/test_gpu_nw_rows.py -q -m gpu -x > gpurun_out/nwc_nw_rows.log 2>&1 || { tail -30 gpurun_out/nwc_nw_rows.log; exit 1; }
tail -2 gpurun_out/nwc_nw_rows.log
python -m pytest tests -q -m gpu -x > gpurun_out/nwc_full.log 2>&1 || { tail -30 gpurun_out/nwc_full.log; exit 1; }
tail -2 gpurun_out/nwc_full.log
python3 bench.py > gpurun_out/nwc_default_bench.json 2> gpurun_out/nwc_default_bench.err
python3 bench.py --reads 4000000 --read-len 150 --cpu-seconds 0 --abi-reads 0 > gpurun_out/nwc_150bp_bench.json 2> gpurun_out/nwc_150bp_bench.err
python3 -c "
import json
for f in ['nwc_default_bench','nwc_150bp_bench']:
    j=json.loads(open('gpurun_out/'+f+'.json').read().strip().splitlines()[-1]); print(f, round(j['value']/1e6,1), j['ms_per_step'], {k:v['ms_per_step'] for k,v in j['kernels'].items()}, j['roofline']['frac'], j['parity_sample']['mismatches'], (j.get('parity_reference') or {}).get('n'), (j.get('parity_reference') or {}).get('mismatches'), j.get('abi_reads_per_s'))
"
