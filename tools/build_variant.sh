#!/bin/bash
# usage: tools/build_variant.sh NAME -DKNOB=value ...   builds raytracing_amd/variants/librt_gpu_NAME.so with extra compiler flags
# (run here, before the GPU call; tools/variants.sh then times the variants on the box)
name=$1; shift
cd "$(dirname "$0")/../raytracing_amd/csrc" || exit 1
mkdir -p ../variants
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt \
  -fno-fast-math -fno-slp-vectorize -Wno-unused-function -I../../include -I. "$@" -shared rt_gpu.hip rt_gpu_multi.hip rt_flatten.cpp \
  -o ../variants/librt_gpu_$name.so -ldl -lpthread
