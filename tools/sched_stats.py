#!/usr/bin/env python3
"""Print the wave scheduler's step counts and lane participation for one benchmark frame."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from raytracing_amd import gpu, scenes
W = H = 1024; spp = int(os.environ.get("SPP", "16"))
sh = scenes.CONFIGS[os.environ.get("SCENE", "dog_room")]()
g = gpu.GpuScene(sh)
buf = torch.zeros((H, W, 3), dtype=torch.float32, device="cuda")
g.render_region_device(buf.data_ptr(), W, H, (0, 0, W, H), spp, 0, gpu.Options(flags=gpu.FLAG_DETAILED_STATS))
torch.cuda.synchronize()
st = g.stats()
waves = W * H // 64
print({k: st[k] for k in ("rays", "shadow_rays", "rng_draws", "root_tests", "tree_entries", "interior_pops", "leaf_pops", "tri_tests", "tri_hits", "kernel_ms")})
for kind in ("shade", "pop", "tri"):
    steps, lanes = st[kind + "_steps"], st[kind + "_lanes"]
    print(f"{kind}: {steps / waves:.0f} steps per wave, {lanes / max(1, steps):.1f} lanes per step")
