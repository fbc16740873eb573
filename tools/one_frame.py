#!/usr/bin/env python3
"""A few frames of one scene, nothing else (for profilers that sample the kernels): SCENE, SIZE, SPP, FRAMES."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from raytracing_amd import gpu, scenes
W = H = int(os.environ.get("SIZE", "1024")); spp = int(os.environ.get("SPP", "64")); n = int(os.environ.get("FRAMES", "3"))
g = gpu.GpuScene(scenes.CONFIGS[os.environ.get("SCENE", "dog_room")]())
buf = torch.zeros((H, W, 3), dtype=torch.float32, device="cuda")
for k in range(n):
    g.render_region_device(buf.data_ptr(), W, H, (0, 0, W, H), spp, 0, gpu.Options(sample_offset=k * spp))
    torch.cuda.synchronize()
print("checksum", float(buf.double().sum()))
g.close()
