#!/bin/bash
# frames in flight with more hardware queues per process
run() { env "$@" timeout -k 10 200 python3 bench.py --no-cpu-baseline --no-side-measurements --steps 36 --warmup 6 --frames-in-flight $FIF > gpurun_out/overlap.tmp 2>&1 || { tail -3 gpurun_out/overlap.tmp; exit 1; }
  python3 -c "
import json
d=json.loads(open('gpurun_out/overlap.tmp').read().strip().splitlines()[-1])
print('in flight $FIF', '$*', d['ms_per_step'], 'ms/frame', d['value'], 'Mrays/s')"; }
for q in 4 8 16; do for FIF in 4 5 6 8; do
run GPU_MAX_HW_QUEUES=$q
done; done
FIF=6; run GPU_MAX_HW_QUEUES=8 RT_GPU_FRESH_BLOCKS_PER_CU_SHARED=2
FIF=6; run GPU_MAX_HW_QUEUES=8 RT_GPU_FRESH_BLOCKS_PER_CU_SHARED=4
FIF=8; run GPU_MAX_HW_QUEUES=8 RT_GPU_FRESH_BLOCKS_PER_CU_SHARED=2
