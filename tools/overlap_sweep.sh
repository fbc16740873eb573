#!/bin/bash
# frames in flight with smaller persistent grids: do launches of different frames share the SIMDs better?
run() { env "$@" timeout -k 10 200 python3 bench.py --no-cpu-baseline --no-side-measurements --steps 32 --warmup 4 --frames-in-flight $FIF > gpurun_out/overlap.tmp 2>&1 || { tail -3 gpurun_out/overlap.tmp; exit 1; }
  python3 -c "
import json
d=json.loads(open('gpurun_out/overlap.tmp').read().strip().splitlines()[-1])
print('in flight $FIF', '$*', d['ms_per_step'], 'ms/frame', d['value'], 'Mrays/s')"; }
for FIF in 4 5; do
run A=0
run RT_GPU_FRESH_BLOCKS_PER_CU=3
run RT_GPU_FRESH_BLOCKS_PER_CU=3 RT_GPU_TREE_BLOCKS_PER_CU=5
run RT_GPU_FRESH_BLOCKS_PER_CU=3 RT_GPU_TREE_BLOCKS_PER_CU=4
run RT_GPU_FRESH_BLOCKS_PER_CU=3 RT_GPU_RESUME_BLOCKS_PER_CU=3
run RT_GPU_FRESH_BLOCKS_PER_CU=3 RT_GPU_TREE_BLOCKS_PER_CU=4 RT_GPU_RESUME_BLOCKS_PER_CU=3
run RT_GPU_FRESH_BLOCKS_PER_CU=2
done
FIF=4; run A=1; run RT_GPU_FRESH_BLOCKS_PER_CU=3
