#!/bin/bash
# Developer run ON the GPU box: rocprofv3 passes (tools/profile.sh) of the staged trilinear kernel's 16-bit and
# per-ray LOD forms next to the gather forms they replace.  usage: tools/dev_prof_staged_ext.sh OUT
OUT=$1
mkdir -p "$OUT"
export VRC_HIP_LIB=
timeout -k 10 500 bash tools/profile.sh "$OUT/u16" -- python3 tools/dev_bench.py --dtype u16 --kernels 2 3 --filters 1 --steps 5 || exit 1
python3 tools/prof_summary.py "$OUT/u16" vrc_k_raycast_lds "" > "$OUT/summary_u16_staged.txt" 2>&1
python3 tools/prof_summary.py "$OUT/u16" "vrc_k_raycast<" "" > "$OUT/summary_u16_gathers.txt" 2>&1
echo "u16 done"
timeout -k 10 500 bash tools/profile.sh "$OUT/raylod" -- python3 tools/dev_bench.py --volume hash --ray-lod 1.0 --levels 0 1 2 3 --kernels 2 0 --filters 1 --steps 5 || exit 1
python3 tools/prof_summary.py "$OUT/raylod" vrc_k_raycast_lds "" > "$OUT/summary_raylod_staged.txt" 2>&1
python3 tools/prof_summary.py "$OUT/raylod" vrc_k_raycast_raylod "" > "$OUT/summary_raylod_gathers.txt" 2>&1
echo "raylod done"
