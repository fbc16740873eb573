#!/bin/bash
# usage: tools/batch_split_sweep.sh OUT  -- one frame alone, cut into 1 / 2 / 4 batches that run on the handle's lanes side by side
cd "$GRAFT_REPO_ROOT" || exit 1
for s in dog_room dogs_and_cat; do for lanes in 3 4; do for paths in 83886080 33554432 22369621 16777216; do
  echo "== $s lanes $lanes batch paths $paths"
  SCENE=$s RT_GPU_STREAMS=$lanes RT_GPU_BATCH_PATHS=$paths PLAIN_ONLY=1 timeout -k 10 120 python3 -u tools/frame_stats.py 2>&1 | grep -v amdgpu.ids || exit 1
done; done; done > gpurun_out/$1 2>&1
cat gpurun_out/$1
