#!/bin/bash
# usage: tools/round_trace.sh   -- durations of every launch of one benchmark frame, in order (rocprofv3 kernel trace), and the rounds' record counts
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
out=gpurun_out/round_trace; rm -rf "$out"; mkdir -p "$out"
RT_GPU_TRACE_ROUNDS=1 PLAIN_ONLY=1 FRAMES=1 timeout -k 10 120 python3 tools/frame_stats.py 2>&1 | grep -v amdgpu.ids | tail -12 > "$out/rounds.txt"
PLAIN_ONLY=1 FRAMES=1 timeout -k 10 200 rocprofv3 --kernel-trace -d "$out/t" -o tr --output-format csv -- python3 tools/frame_stats.py > "$out/trace.log" 2>&1 || { tail -5 "$out/trace.log"; exit 1; }
python3 - "$out" <<'PY'
import csv, glob, sys
out = sys.argv[1]
rows = list(csv.DictReader(open(glob.glob(out + "/t/**/*kernel_trace.csv", recursive=True)[0])))
rows = [r for r in rows if "rt_" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# the last frame: from the last rt_shade_kernel<true, ...> on
start = max(i for i, r in enumerate(rows) if "rt_shade_kernel<true" in r["Kernel_Name"])
t0 = int(rows[start]["Start_Timestamp"])
with open(out + "/launches.txt", "w") as f:
    for r in rows[start:]:
        d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        if d < 10: continue
        f.write(f"{(int(r['Start_Timestamp']) - t0) / 1e3:9.1f} us  +{d:8.1f} us  {r['Kernel_Name'].split('(')[0][5:]}\n")
PY
rm -rf "$out/t"
cat "$out/rounds.txt" "$out/launches.txt"
