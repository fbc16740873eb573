#!/usr/bin/env python3
"""Round trace of one rank's share for WORLD ranks (env WORLD, default 8)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from raytracing_amd import gpu, scenes, multi_gpu
W = H = 1024; spp = 64; world = int(os.environ.get("WORLD", "8"))
g = gpu.GpuScene(scenes.dog_room())
rows = multi_gpu.packed_rows(world, H, multi_gpu.BAND_ROWS)
buf = torch.zeros((rows, W, 3), dtype=torch.float32, device="cuda")
opts = gpu.Options(band_rows=multi_gpu.BAND_ROWS, band_rank=3, band_count=world, flags=gpu.FLAG_COMPACT_BANDS)
for rep in range(3):
    if rep == 2: os.environ["RT_GPU_TRACE_ROUNDS"] = "1"
    torch.cuda.synchronize(); t = time.perf_counter()
    g.render_region_device(buf.data_ptr(), W, H, (0, 0, W, H), spp, rep, opts)
    torch.cuda.synchronize(); dt = time.perf_counter() - t
    print(f"rep {rep}: {1e3*dt:.2f} ms")
st = g.stats(); print({k: st[k] for k in ("shade_ms", "tree_ms", "fresh_ms", "rounds")})
