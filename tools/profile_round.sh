#!/bin/bash
# usage: tools/profile_round.sh TAG   -- kernel trace + stats of the bench command, counter passes, full bench line;
# everything lands in gpurun_out/prof_TAG (copy what is to be kept into profiles/)
tag=$1
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
out=gpurun_out/prof_$tag
rm -rf "$out"; mkdir -p "$out"
# frames one at a time: every kernel runs alone, so the trace's durations are the ones bench.py's roofline uses
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d "$out/trace" -o bench --output-format csv -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-side-measurements --frames-in-flight 1 > "$out/trace.log" 2>&1 || { echo trace failed; tail -5 "$out/trace.log"; exit 1; }
grep '^{' "$out/trace.log" > "$out/bench_under_trace.json"
find "$out/trace" -name '*kernel_stats.csv' -exec cp {} "$out/kernel_stats.csv" \;
# the default command (three frames in flight: kernels of different frames share the machine)
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d "$out/trace3" -o bench --output-format csv -- python3 bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-side-measurements > "$out/trace3.log" 2>&1 || { echo trace3 failed; tail -5 "$out/trace3.log"; exit 1; }
grep '^{' "$out/trace3.log" > "$out/bench_under_trace3.json"
find "$out/trace3" -name '*kernel_stats.csv' -exec cp {} "$out/kernel_stats_in_flight3.csv" \;
find "$out" -name '*kernel_trace.csv' -exec rm {} \;
n=0
for counters in "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" \
                "FETCH_SIZE" "WRITE_SIZE" \
                "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum" \
                "TCC_HIT_sum TCC_MISS_sum" \
                "SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_MUL_F32" \
                "SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_CVT SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM"; do
  n=$((n+1))
  # RT_GPU_PARKED_PERCENT=100: one launch that starts paths per frame (the default smaller lists enqueue safety-net passes whose
  # empty launches make_counters_json.py would count as frames); the counters of the launches with work are the same
  RT_GPU_PARKED_PERCENT=100 timeout -s KILL 150 rocprofv3 --pmc $counters -d "$out/pmc_$n" -o pmc --output-format csv -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-side-measurements --frames-in-flight 1 > "$out/pmc_$n.log" 2>&1 || { echo "pmc pass $n ($counters) failed"; tail -5 "$out/pmc_$n.log"; continue; }
  f=$(find "$out/pmc_$n" -name '*counter_collection.csv' | head -1)
  cp "$f" "$out/pmc_$n.csv"
  python3 tools/pmc_summary.py "$f" > "$out/pmc_$n.txt"
  rm -rf "$out/pmc_$n"
done
python3 tools/make_counters_json.py "$tag" "dog_room 1024x1024 64 spp" "$out"/pmc_[0-9].csv > "$out/counters.json" || echo "counters.json failed"
# the bench line below prices its live kernel times against THESE counters (bench.py reads profiles/counters.json)
[ -s "$out/counters.json" ] && cp "$out/counters.json" profiles/counters.json
timeout -k 10 400 python3 -u bench.py > "$out/bench.json" 2> "$out/bench.err" || { echo bench failed; tail -5 "$out/bench.err"; exit 1; }
rm -rf "$out/trace" "$out/trace3"
echo done
