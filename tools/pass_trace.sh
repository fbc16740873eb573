# rounds of one benchmark frame, printed by the library, for a few parked-list budgets
set -e
export PLAIN_ONLY=1 RT_GPU_STREAMS=1 RT_GPU_TRACE_ROUNDS=1 FRAMES=1
for mb in 98304 12000; do
echo "== dog_room PARKED_MB=$mb"
SCENE=dog_room RT_GPU_PARKED_MB=$mb timeout -k 10 120 python3 tools/frame_stats.py 2>&1 | grep -v amdgpu.ids | head -40
done
