#!/usr/bin/env python3
"""Time per benchmark frame of one scene handle: each frame alone (synchronise after every call) and
frames back to back (calls are asynchronous: the next frame's batches queue behind this one's).
usage: frame_time.py [scene] [width] [spp] [frames]   (environment: RT_GPU_STREAMS, RT_GPU_BATCH_PATHS, ...)"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from raytracing_amd import gpu, scenes

name = sys.argv[1] if len(sys.argv) > 1 else "dog_room"
W = H = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
spp = int(sys.argv[3]) if len(sys.argv) > 3 else 64
n = int(sys.argv[4]) if len(sys.argv) > 4 else 10
sh = scenes.CONFIGS[name]()
t0 = time.perf_counter()
g = gpu.GpuScene(sh)
torch.cuda.synchronize()
create_ms = 1e3 * (time.perf_counter() - t0)
buf = torch.zeros((2, H, W, 3), dtype=torch.float32, device="cuda")
stream = torch.cuda.current_stream().cuda_stream
for k in range(2):
    g.render_region_device(buf[k % 2].data_ptr(), W, H, (0, 0, W, H), spp, 0, gpu.Options(sample_offset=1000 * spp), stream)
torch.cuda.synchronize()
alone = []
for k in range(n):
    t0 = time.perf_counter()
    g.render_region_device(buf[0].data_ptr(), W, H, (0, 0, W, H), spp, 0, gpu.Options(sample_offset=k * spp), stream)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    alone.append((1e3 * (time.perf_counter() - t0), 1e3 * (t1 - t0)))
g.reset_stats()
t0 = time.perf_counter()
for k in range(n):
    g.render_region_device(buf[k % 2].data_ptr(), W, H, (0, 0, W, H), spp, 0, gpu.Options(sample_offset=k * spp), stream)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
st = g.stats()
knobs = {k: v for k, v in os.environ.items() if k.startswith("RT_GPU_")}
print(f"{name} {W}x{H} {spp} spp {knobs}: scene create {create_ms:.1f} ms; alone {min(a for a, _ in alone):.2f} ms min, "
      f"{sorted(a for a, _ in alone)[len(alone) // 2]:.2f} median (host call {sorted(b for _, b in alone)[len(alone) // 2]:.2f} ms); "
      f"back to back {1e3 * (t2 - t0) / n:.2f} ms/frame (host enqueue {1e3 * (t1 - t0) / n:.2f}); "
      f"{st['rays'] / n / 1e6:.1f} Mrays/frame, rounds/frame {st['rounds'] / n:.1f}, batches/frame {st['launches'] / n:.1f}", flush=True)
g.close()
