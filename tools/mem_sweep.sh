#!/bin/bash
# one frame alone (one lane) under smaller parked-list budgets and round limits: what unneeded passes and rounds cost
export PLAIN_ONLY=1 RT_GPU_STREAMS=1
for mb in 98304 10700 8600; do for r in 16 12 8; do
echo "== dog_room PARKED_MB=$mb MAX_ROUNDS=$r"
SCENE=dog_room RT_GPU_PARKED_MB=$mb RT_GPU_MAX_ROUNDS=$r timeout -k 10 120 python3 tools/frame_stats.py 2>&1 | grep -v amdgpu.ids || exit 1
done; done
for r in 16 14 12 10; do
echo "== dogs_and_cat MAX_ROUNDS=$r"
SCENE=dogs_and_cat RT_GPU_MAX_ROUNDS=$r timeout -k 10 120 python3 tools/frame_stats.py 2>&1 | grep -v amdgpu.ids || exit 1
done
