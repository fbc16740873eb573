set -e
export PLAIN_ONLY=1 RT_GPU_STREAMS=1
for scene in dog_room dogs_and_cat; do
for mb in 98304 12000 8000 5000 3500; do
echo "== $scene PARKED_MB=$mb"
SCENE=$scene RT_GPU_PARKED_MB=$mb timeout -k 10 120 python3 tools/frame_stats.py
done; done
