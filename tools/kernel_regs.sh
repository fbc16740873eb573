#!/bin/bash
# Register / scratch / occupancy report of every kernel in rt_gpu.hip (compile only, no GPU needed).
cd "$(dirname "$0")/../raytracing_amd/csrc" || exit 1
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt \
  -fno-fast-math -fno-slp-vectorize -I../../include -I. $EXTRA -Rpass-analysis=kernel-resource-usage --cuda-device-only -c rt_gpu.hip -o /tmp/rt_gpu_dev.o 2>&1 |
  grep -E "Function Name|VGPRs:|SGPRs:|ScratchSize|Occupancy" | sed -E 's/.*remark: +//; s/ \[-Rpass.*//; s/^ +//' |
  awk '/Function Name/{if(line)print line; line=$3; next}{line=line" | "$0}END{print line}' | c++filt
