cd "$GRAFT_REPO_ROOT"
for sc in three_dogs dogs_and_cat; do
  SCENE=$sc RT_GPU_TRACE_ROUNDS=1 PLAIN_ONLY=1 FRAMES=1 timeout -k 10 100 python3 tools/frame_stats.py 2>&1 | grep "batch 0" | tail -17
  SCENE=$sc RT_GPU_MAX_ROUNDS=24 PLAIN_ONLY=1 timeout -k 10 100 python3 tools/frame_stats.py 2>&1 | grep plain
  for fb in 65536 131072 524288 1048576; do echo "FINISH_BELOW=$fb"; SCENE=$sc RT_GPU_MAX_ROUNDS=24 RT_GPU_FINISH_BELOW=$fb PLAIN_ONLY=1 timeout -k 10 100 python3 tools/frame_stats.py 2>&1 | grep plain; done
done
