#!/bin/bash
# frame time per finishing threshold (records) and per tail-lanes setting, three scenes, one frame at a time
export PLAIN_ONLY=1
for s in dog_room three_dogs dogs_and_cat; do
for fb in 262144 524288 1048576 1572864 2097152 3145728; do echo "== $s FINISH_BELOW=$fb"; SCENE=$s RT_GPU_FINISH_BELOW=$fb timeout -k 10 120 python3 -u tools/frame_stats.py 2>&1 | grep -E "back to back|plain" || exit 1; done
for tl in 0 16 48 64; do echo "== $s TAIL_LANES=$tl"; SCENE=$s RT_GPU_TAIL_LANES=$tl timeout -k 10 120 python3 -u tools/frame_stats.py 2>&1 | grep -E "back to back|plain" || exit 1; done
done
