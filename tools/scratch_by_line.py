#!/usr/bin/env python3
"""Where a kernel's scratch (spill) accesses come from: counts of scratch_load/store per source line.
usage: scratch_by_line.py listing.s KERNEL_SUBSTRING   (listing from hipcc -S -g1 --cuda-device-only)"""
import re, collections, sys, os
lines = open(sys.argv[1]).read().split('\n')
want = sys.argv[2]
root = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'raytracing_amd', 'csrc')
src = {n: open(os.path.join(root, n)).read().split('\n') for n in ('rt_trace.hpp', 'rt_gpu.hip', 'rt_math_portable.h')}
files = {}; inside = None; loc = None
cnt = collections.Counter()
for l in lines:
    m = re.match(r'\s*\.file\s+(\d+)\s+"([^"]*)"(?:\s+"([^"]*)")?', l)
    if m: files[m.group(1)] = (m.group(3) or m.group(2)).split('/')[-1]; continue
    m = re.match(r'^(_Z\w+):', l)
    if m: inside = m.group(1); continue
    if l.startswith('.Lfunc_end'): inside = None
    if not inside or want not in inside: continue
    m = re.match(r'\s*\.loc\s+(\d+)\s+(\d+)', l)
    if m: loc = (files.get(m.group(1), m.group(1)), int(m.group(2))); continue
    t = l.strip().split()
    if t and t[0].startswith('scratch_'): cnt[(t[0].split('_')[1], loc)] += 1
print(sum(cnt.values()), 'scratch instructions')
for (op, loc), n in sorted(cnt.items(), key=lambda x: -x[1])[:40]:
    text = src[loc[0]][loc[1] - 1].strip()[:100] if loc and loc[0] in src and loc[1] > 0 else ''
    print(f'{n:3d} {op:5s} {loc[0]}:{loc[1]:<5d} | {text}')
