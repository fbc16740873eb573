set -o pipefail
timeout -k 10 600 python -u -m pytest tests/test_gpu_parity.py -x -q --timeout 240 --timeout-method thread -k "lists_ or render_region_equals_oracle or config5 or several_frames" > gpurun_out/r3d_tests.log 2>&1 || { tail -30 gpurun_out/r3d_tests.log; exit 1; }
tail -3 gpurun_out/r3d_tests.log
for s in dog_room dogs_and_cat three_dogs; do for pc in 100 40; do echo "== $s PARKED_PERCENT=$pc"; SCENE=$s RT_GPU_PARKED_PERCENT=$pc PLAIN_ONLY=1 timeout -k 10 120 python3 -u tools/frame_stats.py 2>&1 | grep -v amdgpu.ids || exit 1; done; done > gpurun_out/r3d_percent.txt 2>&1
cat gpurun_out/r3d_percent.txt
timeout -k 10 300 python3 bench.py --no-cpu-baseline --steps 20 > gpurun_out/r3d_bench.json 2> gpurun_out/r3d_bench.err || { tail gpurun_out/r3d_bench.err; exit 1; }
python3 -c "
import json; d=json.loads(open('gpurun_out/r3d_bench.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['ms_per_frame_single'], d['config']['work_memory_bytes'], d['config']['passes_per_batch'], d['config']['passes_enqueued_per_batch'])"
