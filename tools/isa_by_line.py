#!/usr/bin/env python3
"""Static VALU/SALU instruction count per source line of one kernel in a `hipcc -S -g1` listing.
usage: isa_by_line.py listing.s KERNEL_SUBSTRING [min_count]"""
import collections, re, sys
lines = open(sys.argv[1]).read().split("\n")
want, floor = sys.argv[2], int(sys.argv[3]) if len(sys.argv) > 3 else 8
files, inside, loc = {}, False, ("?", 0)
valu, salu = collections.Counter(), collections.Counter()
for l in lines:
    m = re.match(r'\s*\.file\s+(\d+)\s+"([^"]*)"(?:\s+"([^"]*)")?', l)
    if m:
        files[m.group(1)] = (m.group(3) or m.group(2)).split("/")[-1]; continue
    m = re.match(r"^(_Z\w+):", l)
    if m:
        inside = want in m.group(1); continue
    if l.startswith(".Lfunc_end"): inside = False
    if not inside: continue
    m = re.match(r"\s*\.loc\s+(\d+)\s+(\d+)", l)
    if m:
        loc = (files.get(m.group(1), m.group(1)), int(m.group(2))); continue
    t = l.strip().split()
    if not l.startswith("\t") or not t or t[0][0] in ".;": continue
    if t[0].startswith("v_"): valu[loc] += 1
    elif t[0].startswith("s_"): salu[loc] += 1
tot_v, tot_s = sum(valu.values()), sum(salu.values())
print(f"total VALU {tot_v} SALU {tot_s}")
for k in sorted(set(valu) | set(salu)):
    if valu[k] + salu[k] >= floor:
        print(f"{k[0]}:{k[1]:5d}  valu {valu[k]:4d}  salu {salu[k]:4d}")
