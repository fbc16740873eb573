#!/bin/bash
# usage: tools/keep_profiles.sh TAG   -- copy what tools/profile_round.sh TAG left in gpurun_out/prof_TAG into profiles/ (tracked)
tag=$1; src=gpurun_out/prof_$tag
cd "$(dirname "$0")/.." || exit 1
cp "$src/kernel_stats.csv" "profiles/${tag}_kernel_stats.csv"
cp "$src/kernel_stats_in_flight3.csv" "profiles/${tag}_kernel_stats_in_flight3.csv"
cp "$src/bench.json" "profiles/${tag}_bench.json"
cp "$src/bench_under_trace.json" "profiles/${tag}_bench_under_trace.json"
cp "$src/counters.json" profiles/counters.json
for f in "$src"/pmc_[0-9].csv; do n=$(basename "$f" .csv); gzip -9 -c "$f" > "profiles/${tag}_${n}.csv.gz"; cp "$src/$n.txt" "profiles/${tag}_${n}.txt"; done
ls -la profiles | tail -30
