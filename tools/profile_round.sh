#!/bin/bash
# usage: tools/profile_round.sh TAG   -- kernel trace + stats of the default bench command, HBM traffic passes, full bench line
tag=$1
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
out=gpurun_out/prof_$tag
rm -rf "$out"; mkdir -p "$out"
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d "$out/trace" -o bench --output-format csv -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline > "$out/trace.log" 2>&1 || { echo trace failed; tail -5 "$out/trace.log"; exit 1; }
grep '^{' "$out/trace.log" > "$out/bench_under_trace.json"
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -s KILL 120 rocprofv3 --pmc $c -d "$out/pmc_$c" -o pmc --output-format csv -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline > "$out/pmc_$c.log" 2>&1 || { echo "pmc $c failed"; tail -5 "$out/pmc_$c.log"; exit 1; }
  f=$(find "$out/pmc_$c" -name '*counter_collection.csv' | head -1)
  python3 tools/pmc_summary.py "$f" > "$out/pmc_$c.txt"
done
timeout -k 10 400 python3 -u bench.py > "$out/bench.json" 2> "$out/bench.err" || { echo bench failed; tail -5 "$out/bench.err"; exit 1; }
find "$out/trace" -name '*kernel_stats.csv' -exec cp {} "$out/kernel_stats.csv" \;
find "$out/trace" -name '*kernel_trace.csv' -exec rm {} \;
echo done
