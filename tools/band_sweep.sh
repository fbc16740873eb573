#!/bin/bash
# usage: tools/band_sweep.sh WORLD ENVVAR v1 v2 ...  -- time of one rank's share of the frame for each value
world=$1; var=$2; shift; shift
for v in "$@"; do
  echo -n "WORLD=$world $var=$v  "
  env WORLD=$world $var=$v timeout -k 10 120 python3 -u tools/band_trace.py 2>/dev/null | grep "rep 1" || exit 1
done
