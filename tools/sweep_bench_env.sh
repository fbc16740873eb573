#!/bin/bash
# usage: tools/sweep_bench_env.sh ENVVAR v1 v2 ...   one default bench run (three frames in flight) per value
var=$1; shift
for v in "$@"; do
  env $var=$v timeout -k 10 200 python3 bench.py --no-cpu-baseline --steps 20 > gpurun_out/sweep_bench.tmp 2>&1 || { tail -3 gpurun_out/sweep_bench.tmp; exit 1; }
  python3 -c "
import json,sys
d=json.loads(open('gpurun_out/sweep_bench.tmp').read().strip().splitlines()[-1])
print('$var=$v', d['value'], 'Mrays/s', d['ms_per_step'], 'ms/frame; kernels one at a time', d['roofline']['kernel_ms'], 'ms')
"
done
