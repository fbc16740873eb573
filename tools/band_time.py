#!/usr/bin/env python3
"""Time of one rank's share of the benchmark frame for N = 1, 2, 4, 8 ranks (8-row bands dealt round-robin),
measured on one GPU: what each rank of an N-GPU run computes before the all-gather."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from raytracing_amd import gpu, scenes, multi_gpu
W = H = 1024; spp = 64
g = gpu.GpuScene(scenes.dog_room())
for world in (1, 2, 4, 8):
    rows = multi_gpu.packed_rows(world, H, multi_gpu.BAND_ROWS)
    buf = torch.zeros((rows, W, 3), dtype=torch.float32, device="cuda")
    worst = 0.0
    for rank in range(world):
        opts = gpu.Options(band_rows=multi_gpu.BAND_ROWS, band_rank=rank, band_count=world,
                           flags=gpu.FLAG_COMPACT_BANDS) if world > 1 else gpu.Options()
        for rep in range(3):
            torch.cuda.synchronize(); t = time.perf_counter()
            g.render_region_device(buf.data_ptr(), W, H, (0, 0, W, H), spp, rep, opts)
            torch.cuda.synchronize(); dt = time.perf_counter() - t
        worst = max(worst, dt)
    print(f"ranks {world}: slowest rank {1e3 * worst:.2f} ms  (ideal {37.2 / world:.2f})")
