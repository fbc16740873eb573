#!/usr/bin/env python3
"""Frames per second of ONE scene handle driven by one host thread with F frames in flight: frame k is
ordered on caller stream k % F and uses work-memory lane k % F (options.lane).
usage: inflight_time.py [scene] [width] [spp] [frames] [in_flight ...]"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from raytracing_amd import gpu, scenes

name = sys.argv[1] if len(sys.argv) > 1 else "dog_room"
W = H = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
spp = int(sys.argv[3]) if len(sys.argv) > 3 else 64
n = int(sys.argv[4]) if len(sys.argv) > 4 else 12
flights = [int(x) for x in sys.argv[5:]] or [1, 2, 3]
sh = scenes.CONFIGS[name]()
g = gpu.GpuScene(sh)
for F in flights:
    streams = [torch.cuda.Stream() for _ in range(F)]
    bufs = [torch.zeros((H, W, 3), dtype=torch.float32, device="cuda") for _ in range(F)]

    def run(count, base):
        for k in range(count):
            g.render_region_device(bufs[k % F].data_ptr(), W, H, (0, 0, W, H), spp, 0,
                                   gpu.Options(sample_offset=(base + k) * spp, lane=k % F), streams[k % F].cuda_stream)
        torch.cuda.synchronize()
    run(2 * F, 1000)
    t0 = time.perf_counter()
    run(n, 0)
    dt = time.perf_counter() - t0
    print(f"{name} {W}x{H} {spp} spp, {F} in flight: {1e3 * dt / n:.2f} ms/frame, checksum {float(bufs[0].double().sum()):.6f}", flush=True)
g.close()
