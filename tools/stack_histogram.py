#!/usr/bin/env python3
"""Histogram of the deepest traversal stack per parked ray (rt_gpu_get_stack_histogram) for one frame with the
detailed counters.  environment: SCENE, SIZE, SPP."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from raytracing_amd import gpu, scenes
W = H = int(os.environ.get("SIZE", "1024")); spp = int(os.environ.get("SPP", "16"))
name = os.environ.get("SCENE", "dog_room")
sh = scenes.CONFIGS[name]()
g = gpu.GpuScene(sh)
buf = torch.zeros((H, W, 3), dtype=torch.float32, device="cuda")
g.reset_stats()
g.render_region_device(buf.data_ptr(), W, H, (0, 0, W, H), spp, 0, gpu.Options(flags=gpu.FLAG_DETAILED_STATS))
torch.cuda.synchronize()
hist = g.stack_histogram()
total = sum(hist)
print(f"{name} {W}x{H} {spp} spp: {total} tree walks (rays parked for a tree, tree launches only), stack levels of the scene {g.info()['stack_levels']}")
run = 0
for k, n in enumerate(hist):
    run += n
    if n:
        print(f"  deepest stack {k:2d}: {n:11d}  {100.0 * n / total:7.3f} %   cumulative {100.0 * run / total:8.4f} %")
g.close()
