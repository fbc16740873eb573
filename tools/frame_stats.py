#!/usr/bin/env python3
"""Benchmark frames one at a time with every launch timed: device time per kind of launch, rounds;
then one frame with the detailed counters: work counts and scheduler step occupancy.
environment: SCENE, SIZE, SPP, FRAMES, TREE (sah) and LEAF (triangles per leaf), WORLD (render rank 1's packed share of WORLD ranks); RT_GPU_LIB selects a library variant."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from raytracing_amd import gpu, scenes
W = H = int(os.environ.get("SIZE", "1024")); spp = int(os.environ.get("SPP", "64")); n = int(os.environ.get("FRAMES", "4"))
tree_kw = dict(tree=os.environ["TREE"], leaf_triangles=int(os.environ.get("LEAF", "0"))) if os.environ.get("TREE") else {}
t_assets = time.perf_counter()
sh = scenes.CONFIGS[os.environ.get("SCENE", "dog_room")](**tree_kw)
t_assets = 1e3 * (time.perf_counter() - t_assets)
g = gpu.GpuScene(sh)
if tree_kw:
    from raytracing_amd import host
    print(f"tree {tree_kw}: host assets {t_assets:.0f} ms, " + ", ".join(f"mesh {i}: {host.mesh_census(sh.meshes[i].mesh.contents)}" for i in range(sh.scene.mesh_count)), g.info())
world = int(os.environ.get("WORLD", "1"))
def opts(**kw):
    if world > 1:
        kw["flags"] = kw.get("flags", 0) | gpu.FLAG_COMPACT_BANDS
        kw.update(band_rows=8, band_rank=1, band_count=world)
    return gpu.Options(**kw)
buf = torch.zeros((H, W, 3), dtype=torch.float32, device="cuda")
g.render_region_device(buf.data_ptr(), W, H, (0, 0, W, H), spp, 0, opts(sample_offset=999 * spp))
torch.cuda.synchronize()
g.reset_stats()
wall = []
for k in range(n):
    torch.cuda.synchronize(); t = time.perf_counter()
    g.render_region_device(buf.data_ptr(), W, H, (0, 0, W, H), spp, 0, opts(flags=gpu.FLAG_KERNEL_TIMES, sample_offset=k * spp))
    torch.cuda.synchronize(); wall.append(1e3 * (time.perf_counter() - t))
st, by = g.stats(), g.stats_by_kernel()
torch.cuda.synchronize(); t = time.perf_counter()
for k in range(16):
    g.render_region_device(buf.data_ptr(), W, H, (0, 0, W, H), spp, 0, opts(sample_offset=k * spp))
host = 1e3 * (time.perf_counter() - t) / 16
torch.cuda.synchronize(); back = 1e3 * (time.perf_counter() - t) / 16
print(f"16 calls back to back, no timing events: {back:.2f} ms per call, host enqueue {host:.2f} ms per call")
print(f"plain: wall {sorted(wall)[len(wall) // 2]:.2f} ms median; " + ", ".join(f"{k} {by[k]['kernel_ms'] / n:.2f}" for k in gpu.KERNEL_KINDS) +
      f" ms; sum {st['kernel_ms'] / n:.2f} ms; rounds {st['rounds'] / n:.1f}, batches {st['launches'] / n:.1f}, rays {st['rays'] / n / 1e6:.1f} M, checksum {float(buf.double().sum()):.6f}")
if not os.environ.get("PLAIN_ONLY"):
    g.reset_stats()
    g.render_region_device(buf.data_ptr(), W, H, (0, 0, W, H), spp, 0, opts(flags=gpu.FLAG_DETAILED_STATS))
    torch.cuda.synchronize()
    st = g.stats()
    print({k: st[k] for k in ("tree_entries", "interior_pops", "leaf_pops", "tri_tests", "root_tests")})
    print({k: st[k] for k in ("samples", "rays", "shadow_rays", "rng_draws", "light_tests", "plane_tests", "tri_hits")})
    for kind in ("shade", "pop", "tri"):
        steps, lanes = st[kind + "_steps"], st[kind + "_lanes"]
        print(f"  {kind}: {steps} wave steps, {lanes / max(1, steps):.1f} lanes per step")
g.close()
