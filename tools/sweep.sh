#!/bin/bash
# usage: tools/sweep.sh ENVVAR v1 v2 ...   (runs the bench once per value, prints ms/frame)
var=$1; shift
for v in "$@"; do
  echo -n "$var=$v  "
  env $var=$v timeout -k 10 300 python3 -u bench.py --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], 'ms', d['value'], 'Mrays/s')" || exit 1
done
