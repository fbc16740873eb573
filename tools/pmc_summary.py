#!/usr/bin/env python3
"""Summarise a rocprofv3 --pmc counter_collection.csv per kernel (summed over dispatches)."""
import collections
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(lambda: collections.defaultdict(float))
meta = {}
for r in rows:
    name = r["Kernel_Name"]
    if not name.startswith("void rt_") and "rt_" not in name[:20]:
        continue
    key = name.split("(")[0][-48:]
    agg[key][r["Counter_Name"]] += float(r["Counter_Value"])
    meta[key] = (r["VGPR_Count"], r["LDS_Block_Size"])
    agg[key]["_dispatches"] = len({x["Dispatch_Id"] for x in rows if x["Kernel_Name"] == name})
for k in sorted(agg):
    v = agg[k]
    print(k, "vgpr/lds", meta[k], "dispatches", int(v["_dispatches"]))
    print("   ", {a: f"{b:.4g}" for a, b in sorted(v.items()) if not a.startswith("_")})
    if "SQ_THREAD_CYCLES_VALU" in v and v.get("SQ_ACTIVE_INST_VALU"):
        print("    lanes busy per VALU instruction: %.1f of 64" % (v["SQ_THREAD_CYCLES_VALU"] / v["SQ_ACTIVE_INST_VALU"]))
    if v.get("SQ_WAVE_CYCLES"):
        for c in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_LDS",
                  "SQ_ACTIVE_INST_VMEM", "SQ_ACTIVE_INST_SCA"):
            if c in v:
                print(f"    {c}/WAVE_CYCLES = {v[c] / v['SQ_WAVE_CYCLES']:.3f}")
    if v.get("SQ_BUSY_CYCLES") and v.get("SQ_WAVE_CYCLES"):
        print(f"    avg resident waves (WAVE_CYCLES/BUSY_CYCLES x ...) raw ratio = {v['SQ_WAVE_CYCLES'] / v['SQ_BUSY_CYCLES']:.2f}")
