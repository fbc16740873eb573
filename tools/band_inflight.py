#!/usr/bin/env python3
"""Time per frame of one rank's share of the frame (WORLD ranks, default 8) with K calls in flight, K = 1..6,
FRAMES frames per call (default 1)."""
import os, sys, time
from concurrent.futures import ThreadPoolExecutor
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from raytracing_amd import gpu, scenes, multi_gpu
W = H = 1024; spp = 64; world = int(os.environ.get("WORLD", "8")); per_call = int(os.environ.get("FRAMES", "1"))
sh = scenes.dog_room()
rows = multi_gpu.packed_rows(world, H, multi_gpu.BAND_ROWS)
def options(first):
    o = gpu.Options(band_rows=multi_gpu.BAND_ROWS, band_rank=1, band_count=world, flags=gpu.FLAG_COMPACT_BANDS) if world > 1 else gpu.Options()
    o.frames = per_call; o.sample_offset = first * spp
    return o
lanes = []
for k in range(6):
    lanes.append((gpu.GpuScene(sh), torch.cuda.Stream(), torch.zeros((per_call, rows, W, 3), dtype=torch.float32, device="cuda")))
def render(k, seed):
    g, s, buf = lanes[k]
    g.render_region_device(buf.data_ptr(), W, H, (0, 0, W, H), spp, 0, options(seed * per_call), s.cuda_stream)
for k in range(6): render(k, 100 + k)
torch.cuda.synchronize()
for K in (1, 2, 3, 4, 5, 6):
    pool = ThreadPoolExecutor(max_workers=K)
    n = 24
    torch.cuda.synchronize(); t = time.perf_counter()
    futs = [pool.submit(render, i % K, i) for i in range(K)]
    for i in range(n):
        futs[i].result()
        if i + K < n: futs.append(pool.submit(render, (i + K) % K, i + K))
    torch.cuda.synchronize(); dt = time.perf_counter() - t
    pool.shutdown()
    print(f"world {world}: {per_call} frames per call, {K} calls in flight: {1e3 * dt / n / per_call:.2f} ms per frame share")
