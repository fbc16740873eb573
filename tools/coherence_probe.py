#!/usr/bin/env python3
"""How much does ray order matter to the tree kernel?  The same closest-hit probes (rt_gpu_find_ray_hit: the render
path's tree kernel) in random order and sorted by (instance aimed at, direction octant, Morton code of the origin).
Run under `rocprofv3 --kernel-trace --stats` and compare the rt_tree_kernel durations of the two calls (printed order:
shuffled first, sorted second).  usage: coherence_probe.py [scene] [million rays]"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
from raytracing_amd import gpu, scenes

name = sys.argv[1] if len(sys.argv) > 1 else "dogs_and_cat"
n = int(float(sys.argv[2]) * 1e6) if len(sys.argv) > 2 else 4_000_000
sh = scenes.CONFIGS[name]()
g = gpu.GpuScene(sh)
rng = np.random.default_rng(1)
# diffuse-bounce-like rays: origins on a shell around the meshes' positions, directions toward a jittered mesh point
centres = np.array([[sh.meshes[i].transform.position.x, sh.meshes[i].transform.position.y + 10.0, sh.meshes[i].transform.position.z]
                    for i in range(sh.scene.mesh_count)], np.float32)
which = rng.integers(0, len(centres), n)
o = (centres[which] + rng.normal(0, 25.0, (n, 3))).astype(np.float32)
o[:, 1] = np.abs(o[:, 1])
target = centres[which] + rng.normal(0, 8.0, (n, 3)).astype(np.float32)
d = target - o
d /= np.linalg.norm(d, axis=1, keepdims=True)
rays = np.hstack([o, d]).astype(np.float32)
lo, hi = o.min(0), o.max(0)
cell = np.clip(((o - lo) / (hi - lo + 1e-6) * 32).astype(np.uint32), 0, 31)
def spread(v):
    out = np.zeros_like(v)
    for b in range(5):
        out |= ((v >> b) & 1) << (3 * b)
    return out
morton = spread(cell[:, 0]) | (spread(cell[:, 1]) << 1) | (spread(cell[:, 2]) << 2)
octant = ((d[:, 0] < 0).astype(np.uint32) | ((d[:, 1] < 0).astype(np.uint32) << 1) | ((d[:, 2] < 0).astype(np.uint32) << 2))
key = (which.astype(np.uint64) << 20) | (octant.astype(np.uint64) << 15) | morton.astype(np.uint64)
order = np.argsort(key, kind="stable")
for label, r in (("shuffled", rays), ("sorted", np.ascontiguousarray(rays[order]))):
    g.reset_stats()
    t = time.perf_counter()
    res = g.find_ray_hit(r, 3.4e38)
    dt = time.perf_counter() - t
    st = g.stats()
    print(f"{label}: {n / 1e6:.1f} M rays, call {1e3 * dt:.0f} ms, interior pops {st['interior_pops'] / n:.1f}/ray, tri tests {st['tri_tests'] / n:.1f}/ray", flush=True)
g.close()
