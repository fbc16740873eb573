#!/bin/bash
# usage: tools/other_configs.sh OUT -- the benchmark's line for configurations 1, 2 and 3 at their own sizes (not the headline)
cd "$GRAFT_REPO_ROOT" || exit 1
run() { name=$1; shift; timeout -k 10 200 python bench.py "$@" $BASE --no-side-measurements > gpurun_out/$OUT.$name.json 2> gpurun_out/$OUT.$name.err || { tail -c 300 gpurun_out/$OUT.$name.err; exit 1; }
  python3 -c "
import json; d=json.load(open('gpurun_out/$OUT.$name.json')); print('$name', d['metric'], '|', round(d['value'],1), d['unit'], round(d['ms_per_step'],3), 'ms per frame,', d['config']['ms_per_frame_single'] if 'ms_per_frame_single' in d['config'] else '', 'alone;', 'reference on', d.get('cpu_baseline', {}).get('cores'), 'threads:', d.get('cpu_baseline', {}).get('value'), d.get('cpu_baseline', {}).get('unit'))"; }
OUT=$1
BASE=--no-cpu-baseline; [ -n "$CPU" ] && BASE=   # CPU=1: with the reference timed on the host cores (a bounded sample, about 15 s per configuration)
run config1 --scene deux_spheres --width 512 --height 512 --spp 1 --steps 200 --warmup 20 &&
run config2 --scene sphere_room --width 512 --height 512 --spp 32 --steps 60 --warmup 10 &&
run config3 --scene diffuse_room --width 1024 --height 1024 --spp 256 --max-segments 5 --steps 12 --warmup 3
