#!/usr/bin/env python3
"""Summarise a rocprofv3 --pmc counter_collection.csv for the render kernels."""
import collections
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(dict)
for r in rows:
    if "rt_render" in r["Kernel_Name"]:
        key = (int(r["Dispatch_Id"]), r["Kernel_Name"][:44], r["VGPR_Count"], r["LDS_Block_Size"])
        agg[key][r["Counter_Name"]] = agg[key].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
for k in sorted(agg):
    v = agg[k]
    print(k)
    print("   ", {a: f"{b:.4g}" for a, b in sorted(v.items())})
    if "SQ_THREAD_CYCLES_VALU" in v and v.get("SQ_ACTIVE_INST_VALU"):
        print("    lanes busy per VALU instruction: %.1f of 64" % (v["SQ_THREAD_CYCLES_VALU"] / v["SQ_ACTIVE_INST_VALU"]))
    if "SQ_WAVE_CYCLES" in v:
        for c in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_LDS",
                  "SQ_ACTIVE_INST_VMEM", "SQ_ACTIVE_INST_SCA"):
            if c in v:
                print(f"    {c}/WAVE_CYCLES = {v[c] / v['SQ_WAVE_CYCLES']:.3f}")
