set -e
python3 tools/env_sweep.py --sets "" "" 2>&1 | grep sweep | sed 's/^/[new ] /'
cp gnumap_amd/libgnumap_hip.so /tmp/new.so; cp gnumap_amd/libgnumap_prev.so gnumap_amd/libgnumap_hip.so
python3 tools/env_sweep.py --sets "" "" 2>&1 | grep sweep | sed 's/^/[prev] /'
cp /tmp/new.so gnumap_amd/libgnumap_hip.so
