#!/bin/bash
# One plain-frame line per scene and per threshold of the finishing launch (parked records at or below which the rounds end).
cd "$GRAFT_REPO_ROOT" || exit 1
for sc in dog_room three_dogs dogs_and_cat; do
  for fb in 131072 262144 524288 1048576 2097152 4194304; do
    echo -n "$sc FINISH_BELOW=$fb  "
    SCENE=$sc RT_GPU_FINISH_BELOW=$fb PLAIN_ONLY=1 timeout -k 10 100 python3 tools/frame_stats.py 2>&1 | grep plain || exit 1
  done
done
