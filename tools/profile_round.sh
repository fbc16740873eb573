#!/bin/bash
# usage: tools/profile_round.sh TAG   -- kernel trace + stats of the default bench command, HBM traffic passes, full bench line
tag=$1
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
out=gpurun_out/prof_$tag
rm -rf "$out"; mkdir -p "$out"
# frames one at a time: every kernel runs alone, so the trace's durations are the ones bench.py's roofline uses
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d "$out/trace" -o bench --output-format csv -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --frames-in-flight 1 > "$out/trace.log" 2>&1 || { echo trace failed; tail -5 "$out/trace.log"; exit 1; }
grep '^{' "$out/trace.log" > "$out/bench_under_trace.json"
# the default command (three frames in flight: kernels of different frames share the machine)
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d "$out/trace3" -o bench --output-format csv -- python3 bench.py --steps 6 --warmup 3 --no-cpu-baseline > "$out/trace3.log" 2>&1 || { echo trace3 failed; tail -5 "$out/trace3.log"; exit 1; }
grep '^{' "$out/trace3.log" > "$out/bench_under_trace3.json"
find "$out/trace3" -name '*kernel_stats.csv' -exec cp {} "$out/kernel_stats_in_flight3.csv" \;
find "$out/trace3" -name '*kernel_trace.csv' -exec rm {} \;
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -s KILL 120 rocprofv3 --pmc $c -d "$out/pmc_$c" -o pmc --output-format csv -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --frames-in-flight 1 > "$out/pmc_$c.log" 2>&1 || { echo "pmc $c failed"; tail -5 "$out/pmc_$c.log"; exit 1; }
  f=$(find "$out/pmc_$c" -name '*counter_collection.csv' | head -1)
  python3 tools/pmc_summary.py "$f" > "$out/pmc_$c.txt"
done
timeout -k 10 400 python3 -u bench.py > "$out/bench.json" 2> "$out/bench.err" || { echo bench failed; tail -5 "$out/bench.err"; exit 1; }
find "$out/trace" -name '*kernel_stats.csv' -exec cp {} "$out/kernel_stats.csv" \;
find "$out/trace" -name '*kernel_trace.csv' -exec rm {} \;
echo done
