#!/bin/bash
# profiles/isa_cycles.json: static cycles per vector instruction of the render kernels (no GPU needed); bench.py prices the
# counted instructions of each kind of launch with it (roofline.kernels.*.valu_cycle_frac_estimate)
cd "$(dirname "$0")/../raytracing_amd/csrc" || exit 1
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt \
  -fno-fast-math -fno-slp-vectorize -I../../include -I. --cuda-device-only -S rt_gpu.hip -o /tmp/rt_gpu_cycles.s 2>/dev/null || exit 1
python3 ../../tools/isa_cycles.py --json /tmp/rt_gpu_cycles.s > ../../profiles/isa_cycles.json
