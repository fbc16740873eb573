#!/usr/bin/env python3
"""profiles/counters.json from rocprofv3 --pmc passes of
`bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-side-measurements --frames-in-flight 1`
(one pass per counter group; tools/profile_round.sh runs them).  That command renders several plain
frames and one with the detailed counters (kernel instantiations <..., true>, left out here); every
counter is summed over all launches of a kind and divided by the number of plain frames (= launches
of rt_shade_kernel<true, false, ...>, one per frame).
usage: make_counters_json.py TAG WORKLOAD sq.csv fetch.csv write.csv [more.csv ...] > profiles/counters.json"""
import collections, csv, json, sys

tag, workload, paths = sys.argv[1], sys.argv[2], sys.argv[3:]


def kind_of(name):
    key = name.split("(")[0].replace("void ", "")
    if not key.startswith("rt_") or "<" not in key or ">" not in key:
        return None
    args = [a.strip() for a in key[key.index("<") + 1:key.rindex(">")].split(",")]
    if key.startswith("rt_shade_kernel"):
        if args[1] == "true":
            return None                                  # the frame with detailed counters
        return "new_paths" if args[0] == "true" else "resumed_paths"
    if key.startswith("rt_tree_kernel"):
        return None if args[0] == "true" else "tree"
    if key.startswith("rt_finish_kernel"):
        return None if args[0] == "true" else "finish"
    return None


sums = collections.defaultdict(lambda: collections.defaultdict(float))
names = collections.defaultdict(set)
frames = 0
for path in paths:
    seen = set()
    for r in csv.DictReader(open(path)):
        k = kind_of(r["Kernel_Name"])
        if not k:
            continue
        sums[k][r["Counter_Name"]] += float(r["Counter_Value"])
        names[k].add(r["Kernel_Name"].split("(")[0].replace("void ", "").strip())
        if k == "new_paths":
            seen.add(r["Dispatch_Id"])
    frames = frames or len(seen)
    assert len(seen) == frames, (path, len(seen), frames)

kinds = {}
for k, c in sums.items():
    e = {"valu_wave_instructions": c.get("SQ_INSTS_VALU", 0.0) / frames,
         "fetch_bytes_raw": c.get("FETCH_SIZE", 0.0) * 1024.0 / frames,
         "write_bytes": c.get("WRITE_SIZE", 0.0) * 1024.0 / frames}
    if c.get("SQ_ACTIVE_INST_VALU"):
        e["lanes_active"] = round(c["SQ_THREAD_CYCLES_VALU"] / c["SQ_ACTIVE_INST_VALU"], 1)
    if c.get("SQ_WAVE_CYCLES"):
        for name in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU"):
            if name in c:
                e[name.lower() + "_per_wave_cycle"] = round(c[name] / c["SQ_WAVE_CYCLES"], 3)
    for name in sorted(c):
        if name.startswith(("TCP_", "TCC_", "SQ_INSTS_", "SQ_INST_")) and name != "SQ_INSTS_VALU":
            e[name] = c[name] / frames
    e["kernel_names"] = sorted(names[k])
    kinds[k] = e
json.dump({
    "workload": workload,
    "frames_averaged": frames,
    "source": f"profiles/{tag}_pmc_*.csv.gz (raw) and .txt (per-kernel sums): rocprofv3 --pmc, one pass per counter group, of `python3 bench.py --steps 1 --warmup 0 "
              "--no-cpu-baseline --no-side-measurements --frames-in-flight 1` (the mesh-heavy scene: of `python3 tools/one_frame.py`), RT_GPU_PARKED_PERCENT=100; per frame and kind of launch; FETCH_SIZE and WRITE_SIZE in KiB x 1024, raw "
              "(the read counter tallies 128-byte requests at 64 bytes on gfx950: bench.py doubles it)",
    "kinds": kinds,
}, sys.stdout, indent=1)
print()
