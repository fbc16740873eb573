#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
out=gpurun_out/coherence; rm -rf "$out"; mkdir -p "$out"
for sc in dog_room dogs_and_cat; do
  timeout -k 10 250 rocprofv3 --kernel-trace -d "$out/t_$sc" -o tr --output-format csv -- python3 tools/coherence_probe.py $sc 4 > "$out/$sc.log" 2>&1 || { tail -5 "$out/$sc.log"; exit 1; }
  grep -E "shuffled|sorted" "$out/$sc.log"
  python3 - "$out/t_$sc" <<'PY'
import csv, glob, sys
rows = list(csv.DictReader(open(glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0])))
rows = [r for r in rows if "rt_tree_kernel" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
print("rt_tree_kernel durations (ms):", [round((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6, 3) for r in rows])
PY
  rm -rf "$out/t_$sc"
done
