#!/usr/bin/env python3
"""Static instruction census per kernel of a hipcc -S listing (classes: VALU, fp64, division pieces, SALU, VMEM, LDS)."""
import collections, re, sys
lines = open(sys.argv[1]).read().split("\n")
name, c = None, None
def flush():
    if name: print(f"{name[:44]:44s}", dict(c))
for l in lines:
    m = re.match(r"^(_Z\w+):", l)
    if m:
        flush(); name, c = m.group(1), collections.Counter(); continue
    if l.startswith(".Lfunc_end"):
        flush(); name = None; continue
    if name is None or not l.startswith("\t"): continue
    t = l.strip().split()
    if not t or t[0][0] in ".;": continue
    x = t[0]; c["all"] += 1
    if x.startswith("v_"):
        c["valu"] += 1
        if "f64" in x: c["f64"] += 1
        if x.startswith(("v_div_", "v_rcp")): c["div_part"] += 1
        if x.startswith("v_rcp_f32"): c["rcp32"] += 1
        if x.startswith("v_sqrt"): c["sqrt"] += 1
        if x.startswith(("v_cndmask", "v_mov")): c["mov_sel"] += 1
        if x.startswith("v_cmp"): c["cmp"] += 1
    elif x.startswith("s_"):
        c["salu"] += 1
        if x.startswith("s_load"): c["smem"] += 1
        if x.startswith("s_waitcnt"): c["wait"] += 1
        if x.startswith(("s_cbranch", "s_branch")): c["branch"] += 1
    elif x.startswith(("global_", "buffer_", "flat_", "scratch_")): c["vmem"] += 1
    elif x.startswith("ds_"): c["lds"] += 1
flush()
