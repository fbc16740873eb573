#!/usr/bin/env python3
"""One frame with the detailed counters: wave steps and lanes per step of each kind of launch (SCENE, SIZE, SPP)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from raytracing_amd import gpu, scenes
W = H = int(os.environ.get("SIZE", "1024")); spp = int(os.environ.get("SPP", "64"))
g = gpu.GpuScene(scenes.CONFIGS[os.environ.get("SCENE", "dog_room")]())
buf = torch.zeros((H, W, 3), dtype=torch.float32, device="cuda")
g.render_region_device(buf.data_ptr(), W, H, (0, 0, W, H), spp, 0)
torch.cuda.synchronize(); g.reset_stats()
g.render_region_device(buf.data_ptr(), W, H, (0, 0, W, H), spp, 0, gpu.Options(flags=gpu.FLAG_DETAILED_STATS))
torch.cuda.synchronize()
by = g.stats_by_kernel()
for k in gpu.KERNEL_KINDS:
    s = by[k]
    print(k, {n: s[n] for n in ("rays", "shadow_rays", "rng_draws", "tri_hits", "plane_tests", "shade_steps", "shade_lanes", "pop_steps", "pop_lanes", "tri_steps", "tri_lanes")},
          f"lanes per shade step {s['shade_lanes'] / max(1, s['shade_steps']):.1f}")
g.close()
