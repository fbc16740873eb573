#!/usr/bin/env python3
"""One-off, larger than the test suite: the five benchmark configurations at their own frame sizes and sample counts
(configuration 5 at 1024 x 1024 x 64 spp: its own 4096 x 4096 x 1024 spp would keep the oracle busy for days), GPU against
the oracle, every float of the frame bit for bit.  The oracle renders row bands on a pool of threads (its results do not
depend on how a frame is cut into regions).
usage: python tests/config_parity.py [threads]   (run by hand on a GPU machine; not collected by pytest)"""
import ctypes as C
import os
import sys
import time
from concurrent.futures import ThreadPoolExecutor

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import _libs as L
from raytracing_amd import gpu, host, scenes

threads = int(sys.argv[1]) if len(sys.argv) > 1 else 16
CASES = [("1", "deux_spheres", 512, 1, 64), ("2", "sphere_room", 512, 32, 64), ("3", "diffuse_room", 1024, 256, 5),
         ("4", "dog_room", 1024, 64, 64), ("5 (at 1024 x 1024 x 64 spp)", "dogs_and_cat", 1024, 64, 64)]
bad = 0
for label, name, side, spp, segments in CASES:
    sh = scenes.CONFIGS[name]()
    gs = gpu.GpuScene(sh)
    img, got = host.new_image(side, side)
    gs.render_region(img, (0, 0, side, side), spp, 5, gpu.Options(max_segments=segments))
    t = time.time()
    ref_img, want = host.new_image(side, side)

    def band(y0, rows=16):
        r = gpu.rect(0, y0, side, min(rows, side - y0))
        L.oracle().orc_render_region_counter(sh.ptr, C.byref(ref_img), C.byref(r), spp, 0, 5, L.MATH_PORTABLE, segments, None)
        return y0

    with ThreadPoolExecutor(threads) as pool:
        for k, _ in enumerate(pool.map(band, range(0, side, 16))):
            if k % 8 == 7:
                print(f"  configuration {label}: oracle at row {16 * (k + 1)} of {side}, {time.time() - t:.0f} s", flush=True)
    same = L.same_bits(got, want)
    print(f"configuration {label}: {name} {side}x{side} x {spp} spp, paths of up to {segments} segments: {int((~same).sum())} differing values "
          f"of {same.size}, oracle {time.time() - t:.1f} s on {threads} threads, rays {gs.stats()['rays']}", flush=True)
    bad += int((~same).sum())
    gs.close()
sys.exit(1 if bad else 0)
