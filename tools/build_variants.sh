#!/bin/bash
# Developer A/B builds of libvrc_hip.so with different compile-time knobs.
# usage: tools/build_variants.sh NAME "-DVRC_GROUP=4 ..." ; result: gpurun_variants/libvrc_hip_NAME.so
set -e
cd "$(dirname "$0")/.."
mkdir -p variants
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -std=c++17 -fPIC -shared -fno-slp-vectorize -DVRC_DEV_BUILD $2 -o variants/libvrc_hip_$1.so libre_amd/csrc/vrc_api.hip libre_amd/csrc/vrc_kernels.hip libre_amd/csrc/vrc_kernels_lds.hip libre_amd/csrc/vrc_kernels_raylod.hip libre_amd/csrc/vrc_comm.hip -ldl
