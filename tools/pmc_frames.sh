#!/bin/bash
# usage: tools/pmc_frames.sh TAG "COUNTER COUNTER ..." ["second pass" ...]   -- one rocprofv3 --pmc pass per counter list over
# tools/one_frame.py (plain frames of SCENE, nothing else); per-kernel sums on stdout
tag=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
n=0
for counters in "$@"; do
  n=$((n+1)); out=gpurun_out/pmcf_${tag}_$n
  rm -rf "$out"
  timeout -s KILL 120 rocprofv3 --pmc $counters -d "$out" -o pmc --output-format csv -- python3 tools/one_frame.py > "$out.log" 2>&1 || { echo "pass $n failed"; tail -5 "$out.log"; exit 1; }
  f=$(find "$out" -name '*counter_collection.csv' | head -1)
  echo "== pass $n: $counters"
  python3 tools/pmc_summary.py "$f"
  rm -rf "$out"
done
