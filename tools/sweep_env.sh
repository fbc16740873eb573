#!/bin/bash
# usage: tools/sweep_env.sh ENVVAR v1 v2 ...   one frame_stats run per value (plain frame line only)
var=$1; shift
for v in "$@"; do
  echo -n "$var=$v  "
  env $var=$v timeout -k 10 120 python3 -u tools/frame_stats.py 2>&1 | grep "detailed=False" || exit 1
done
