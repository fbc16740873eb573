#!/bin/bash
# usage: tools/pmc.sh TAG "COUNTER COUNTER ..." ["COUNTERS of a second pass" ...]
# One rocprofv3 --pmc pass per counter list over a one-frame bench run; per-kernel sums on stdout.
tag=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
n=0
for counters in "$@"; do
  n=$((n+1)); out=gpurun_out/pmc_${tag}_$n
  rm -rf "$out"
  timeout -s KILL 120 rocprofv3 --pmc $counters -d "$out" -o pmc --output-format csv -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-side-measurements --frames-in-flight 1 > "$out.log" 2>&1 || { echo "pass $n failed"; tail -5 "$out.log"; exit 1; }
  f=$(find "$out" -name '*counter_collection.csv' | head -1)
  echo "== pass $n: $counters"
  python3 tools/pmc_summary.py "$f"
done
