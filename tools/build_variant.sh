#!/bin/bash
# Builds the library with extra compiler flags into build_variants/librt_<name>.so (git-ignored; it
# travels to the GPU box), for same-box A/B runs:  tools/build_variant.sh stride84 -DRT_LDS_NODE_WORDS=21
# then  python tools/regress.py --lib build_variants/librt_stride84.so
set -e
cd "$(dirname "$0")/../raytrace_amd/csrc"
name=$1; shift
mkdir -p ../../build_variants
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC -shared --offload-arch=gfx950 -fno-gpu-rdc -Wall -Wno-unused-function \
    -ffp-contract=on -fno-slp-vectorize "$@" \
    -x hip rt_api.cpp -x hip rt_bvh.cpp -x c++ rt_image.cpp -x hip rt_kernels.hip -x hip rt_lbvh.hip -x hip rt_reconstruct.hip -o ../../build_variants/librt_$name.so
echo built build_variants/librt_$name.so
