#!/usr/bin/env python3
"""One benchmark frame with the detailed counters: rounds, kernel shares, tree step occupancy."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from raytracing_amd import gpu, scenes
W = H = int(os.environ.get("SIZE", "1024")); spp = int(os.environ.get("SPP", "64"))
sh = scenes.CONFIGS[os.environ.get("SCENE", "dog_room")]()
g = gpu.GpuScene(sh)
buf = torch.zeros((H, W, 3), dtype=torch.float32, device="cuda")
for detailed in (False, True):
    g.reset_stats()
    torch.cuda.synchronize(); t = time.perf_counter()
    g.render_region_device(buf.data_ptr(), W, H, (0, 0, W, H), spp, 0,
                           gpu.Options(flags=gpu.FLAG_DETAILED_STATS if detailed else 0))
    torch.cuda.synchronize(); dt = time.perf_counter() - t
    st = g.stats()
    print(f"detailed={detailed}: wall {1e3*dt:.1f} ms, shade {st['shade_ms']:.1f} ms, tree {st['tree_ms']:.1f} ms, "
          f"rounds {st['rounds']}, overflowed {st['overflowed']}, rays {st['rays']/1e6:.1f} M")
    if detailed:
        print({k: st[k] for k in ("tree_entries", "interior_pops", "leaf_pops", "tri_tests", "root_tests")})
        print({k: st[k] for k in ("samples", "rays", "shadow_rays", "rng_draws", "light_tests", "plane_tests", "tri_hits")})
        for kind in ("shade", "pop", "tri"):
            steps, lanes = st[kind + "_steps"], st[kind + "_lanes"]
            print(f"  {kind}: {steps} wave steps, {lanes / max(1, steps):.1f} lanes per step")
