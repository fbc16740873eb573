#!/bin/bash
# usage: tools/occupancy_pmc.sh lib1.so lib2.so ...  -- for each library variant (tree kernel compiled for another number of waves per
# SIMD) and two scenes: one counter pass over tools/one_frame.py, the tree kernel's sums (waiting shares, vector activity, waves)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
export FRAMES=2
for scene in dog_room dogs_and_cat; do for lib in "$@"; do
  echo "#### $scene $lib"
  out=gpurun_out/occ_pmc
  rm -rf "$out"
  SCENE=$scene RT_GPU_LIB=$PWD/$lib timeout -s KILL 150 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAVES -d "$out" -o pmc --output-format csv -- python3 tools/one_frame.py > "$out.log" 2>&1 || { echo "pass failed"; tail -5 "$out.log"; exit 1; }
  f=$(find "$out" -name '*counter_collection.csv' | head -1)
  python3 tools/pmc_summary.py "$f" | grep -A8 "rt_tree_kernel<false>"
  SCENE=$scene RT_GPU_LIB=$PWD/$lib PLAIN_ONLY=1 timeout -k 10 120 python3 -u tools/frame_stats.py 2>&1 | grep plain
done; done
