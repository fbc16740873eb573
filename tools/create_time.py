import sys, time; sys.path.insert(0,__import__('os').path.join(__import__('os').path.dirname(__import__('os').path.abspath(__file__)), '..'))
import torch
from raytracing_amd import gpu, scenes
for name in ("dog_room","dogs_and_cat"):
    sh = scenes.CONFIGS[name]()
    for k in range(3):
        t=time.perf_counter(); g=gpu.GpuScene(sh); dt=1e3*(time.perf_counter()-t)
        print(name, k, f"{dt:.1f} ms", g.info()); g.close()
