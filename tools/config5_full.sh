#!/bin/bash
# usage: tools/config5_full.sh SPP OUT -- the mesh-heavy scene at 4096 x 4096 with SPP samples per pixel, one frame timed after one
# set-up frame, one call in flight (a heartbeat file keeps the run from looking hung)
cd "$GRAFT_REPO_ROOT" || exit 1
( while sleep 25; do date >> gpurun_out/heartbeat.txt; done ) &
hb=$!
timeout -k 10 ${LIMIT:-420} python bench.py --scene dogs_and_cat --width 4096 --height 4096 --spp $1 --steps 1 --warmup 0 --frames-in-flight 1 --no-cpu-baseline --no-side-measurements > gpurun_out/$2.json 2> gpurun_out/$2.err
rc=$?
kill $hb
tail -c 400 gpurun_out/$2.err
[ $rc -eq 0 ] && python3 -c "
import json,sys; d=json.load(open('gpurun_out/$2.json')); print(d['metric']); print(d['value'], d['unit'], d['ms_per_step'], d['config'])"
exit $rc
