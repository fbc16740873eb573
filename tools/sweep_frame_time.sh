#!/bin/bash
# usage: tools/sweep_frame_time.sh "ENV=val ENV=val" "ENV=val" ...   -- tools/frame_time.py once per environment setting
cd "$GRAFT_REPO_ROOT" || exit 1
for setting in "$@"; do
  env $setting timeout -k 10 120 python3 tools/frame_time.py $FRAME_ARGS || exit 1
done
