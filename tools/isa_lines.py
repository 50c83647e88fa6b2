#!/usr/bin/env python3
"""Static vector / scalar instruction counts of one kernel per source line (hipcc -gline-tables-only -S).
usage: tools/isa_lines.py <file.hip> <mangled-name-prefix> [top N]"""
import collections, re, subprocess, sys, os
src_path, prefix = sys.argv[1], sys.argv[2]
top = int(sys.argv[3]) if len(sys.argv) > 3 else 40
out = "/tmp/isa_lines.s"
subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-gline-tables-only", "-I", os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "include"),
                "-S", "--cuda-device-only", src_path, "-o", out], check=True, stderr=subprocess.DEVNULL)
s = open(out).read().split("\n")
start = [i for i, l in enumerate(s) if l.startswith(prefix)][0]
end = [i for i, l in enumerate(s) if i > start and l.startswith(".Lfunc_end")][0]
files = {}
for l in s:
    m = re.match(r'\s*\.file\s+(\d+)\s+"([^"]*)"(?:\s+"([^"]*)")?', l)
    if m:
        files[int(m.group(1))] = (m.group(3) or m.group(2))
cur = None; cv = collections.Counter(); cs = collections.Counter(); cl = collections.Counter()
for l in s[start:end]:
    t = l.strip()
    m = re.match(r'\.loc\s+(\d+)\s+(\d+)', t)
    if m:
        cur = (files.get(int(m.group(1)), '?').split('/')[-1], int(m.group(2))); continue
    if not t or t.startswith(('.', ';', '//')) or t.endswith(':'):
        continue
    if t.startswith('v_'): cv[cur] += 1
    elif t.startswith('s_'): cs[cur] += 1
    elif t.startswith(('ds_', 'global_', 'flat_', 'buffer_')): cl[cur] += 1
print('static VALU', sum(cv.values()), 'SALU', sum(cs.values()), 'mem', sum(cl.values()))
src = open(src_path).read().split('\n')
for (f, ln), c in sorted(cv.items(), key=lambda x: -x[1])[:top]:
    txt = src[ln - 1].strip()[:120] if f == os.path.basename(src_path) and 0 < ln <= len(src) else ''
    print(f'{c:4d} v {cs[(f, ln)]:4d} s {cl[(f, ln)]:3d} m  {f}:{ln}  {txt}')
