#!/bin/bash
# usage: tools/variants.sh lib1.so lib2.so ...   one frame_stats run per library variant (paths relative to the repo root)
cd "$GRAFT_REPO_ROOT" 2>/dev/null
for lib in "$@"; do
  echo "== $lib"
  RT_GPU_LIB=$PWD/$lib PLAIN_ONLY=1 timeout -k 10 120 python3 -u tools/frame_stats.py 2>&1 | grep -v amdgpu.ids || exit 1
done
