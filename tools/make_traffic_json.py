#!/usr/bin/env python3
"""profiles/hbm_traffic.json from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) of
`bench.py --steps 1 --warmup 0 --no-cpu-baseline --frames-in-flight 1`.  That command renders
several plain frames and one with the detailed counters (kernel instantiations <..., true>); the
plain frames' kernels are summed and divided by the number of plain frames (= dispatches of
the rt_shade_kernel<true, ...> instantiation, one per frame).  usage: make_traffic_json.py FETCH.csv WRITE.csv TAG"""
import csv, json, sys

def per_kernel(path, counter):
    out = {}
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        name = r["Kernel_Name"]
        if "rt_" not in name:
            continue
        key = name.split("(")[0].replace("void ", "")
        args = [a.strip() for a in key[key.index("<") + 1:key.rindex(">")].split(",")] if "<" in key else []
        stats_arg = 1 if key.startswith("rt_shade_kernel") else 0      # position of the detailed-counters flag
        if args and args[stats_arg] == "true":
            continue                                   # the frame with detailed counters
        out.setdefault(key, []).append(float(r["Counter_Value"]) * 1024.0)
    return out

fetch, write = per_kernel(sys.argv[1], "FETCH_SIZE"), per_kernel(sys.argv[2], "WRITE_SIZE")
frames = len(next(v for k, v in fetch.items() if k.startswith("rt_shade_kernel<true,")))
kernels = {}
for k in sorted(set(fetch) | set(write)):
    f, w = fetch.get(k, []), write.get(k, [])
    if k == "rt_resolve_kernel":                       # also runs once in the detailed frame
        f, w = f[:frames], w[:frames]
    kernels[k] = {"launches_per_frame": round(max(len(f), len(w)) / frames, 2),
                  "fetch_bytes": int(sum(f) / frames), "write_bytes": int(sum(w) / frames)}
total = sum(v["fetch_bytes"] + v["write_bytes"] for v in kernels.values())
json.dump({
    "workload": "dog_room 1024x1024 64 spp",
    "bytes_per_frame": int(total),
    "frames_averaged": frames,
    "source": f"profiles/{sys.argv[3]}_pmc_FETCH_SIZE.csv + profiles/{sys.argv[3]}_pmc_WRITE_SIZE.csv: rocprofv3 --pmc FETCH_SIZE / "
              "--pmc WRITE_SIZE (separate passes), all launches of the render kernels of a frame summed, KiB x 1024; raw "
              "counter values (the read counter is known to under-report wide streaming reads by 2x on gfx950, so the "
              "true figure lies between this and this plus the fetch bytes)",
    "kernels": kernels,
}, sys.stdout, indent=1)
print()
