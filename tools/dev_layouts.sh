#!/bin/bash
# Developer experiment: build libvrc_hip.so variants with other atlas layouts / lane orders / tile shapes
# into variants/ (git-ignored, shipped to the GPU box by gpurun).  Only the table-driven point-sampling
# kernel (VRC_KERNEL_GRID_DDA, fixed-point stepping) is meaningful in a VRC_LAYOUT != 0 build.
# usage: tools/dev_layouts.sh name "-DVRC_LAYOUT=2 -DVRC_LANES_ROWMAJOR" [name flags]...
set -e
cd "$(dirname "$0")/.."
mkdir -p variants
S=libre_amd/csrc
while [ $# -ge 2 ]; do
  name=$1; flags=$2; shift; shift
  ( /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -std=c++17 -fPIC -shared -fno-slp-vectorize -DVRC_DEV_BUILD $flags \
      -o variants/$name.so $S/vrc_api.hip $S/vrc_kernels.hip $S/vrc_kernels_lds.hip $S/vrc_kernels_raylod.hip $S/vrc_comm.hip -ldl \
      > variants/$name.log 2>&1 && echo "built $name" || echo "FAILED $name (variants/$name.log)" ) &
  while [ $(jobs -r | wc -l) -ge 6 ]; do sleep 1; done
done
wait
