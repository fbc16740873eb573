#!/usr/bin/env python3
"""One-off, larger than the test suite: whole frames of every scene, GPU against the oracle, bit for bit.
usage: python tests/big_parity.py [side] [spp]   (run by hand on a GPU machine; not collected by pytest)"""
import ctypes as C
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import _libs as L
from raytracing_amd import gpu, host, scenes

side = int(sys.argv[1]) if len(sys.argv) > 1 else 384
spp = int(sys.argv[2]) if len(sys.argv) > 2 else 4
bad = 0
cases = [(name, {}) for name in sorted(scenes.CONFIGS)] + [("dogs_and_cat", dict(tree="sah", leaf_triangles=4))]
for name, kw in cases:
    sh = scenes.CONFIGS[name](**kw)
    if kw:
        name += " (surface-area tree)"
    gs = gpu.GpuScene(sh)
    img, got = host.new_image(side, side)
    gs.render_region(img, (0, 0, side, side), spp, 77)
    t = time.time()
    ref_img, want = host.new_image(side, side)
    r = gpu.rect(0, 0, side, side)
    L.oracle().orc_render_region_counter(sh.ptr, C.byref(ref_img), C.byref(r), spp, 0, 77, L.MATH_PORTABLE, 64, None)
    same = L.same_bits(got, want)
    print(f"{name}: {side}x{side} x {spp} spp, {int((~same).sum())} differing values, oracle {time.time() - t:.1f} s, "
          f"rounds {gs.stats()['rounds']}", flush=True)
    bad += int((~same).sum())
    gs.close()
sys.exit(1 if bad else 0)
