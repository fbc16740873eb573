#!/bin/bash
# one rank's eighth of the benchmark frame, alone: finishing threshold, and the share's samples in several batches on the handle's lanes
cd "$GRAFT_REPO_ROOT" 2>/dev/null
for bp in 8388608 4194304 2796203 2097152 1398102; do
  echo "BATCH_PATHS=$bp"; RT_GPU_BATCH_PATHS=$bp WORLD=8 PLAIN_ONLY=1 timeout -k 10 100 python3 tools/frame_stats.py 2>&1 | grep -E "plain|calls" || exit 1
done
echo "BATCH_PATHS=2796203 STREAMS=4"; RT_GPU_STREAMS=4 RT_GPU_BATCH_PATHS=2097152 WORLD=8 PLAIN_ONLY=1 timeout -k 10 100 python3 tools/frame_stats.py 2>&1 | grep -E "plain|calls" || exit 1
