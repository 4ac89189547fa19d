#!/bin/bash
# Developer run ON the GPU box: rocprofv3 passes (tools/profile.sh) of the LDS-staged trilinear kernel on BASELINE C2
# for the product library and for a reference build (round 2's), condensed into two summaries.
# usage: tools/dev_prof_lds.sh OUT [REFERENCE_LIB]
OUT=$1; REF=$2
mkdir -p "$OUT"
export VRC_HIP_LIB=
timeout -k 10 500 bash tools/profile.sh "$OUT/now" -- python3 tools/dev_bench.py --kernels 3 --filters 1 --steps 5
python3 tools/prof_summary.py "$OUT/now" vrc_k_raycast_lds "" > "$OUT/summary_now.txt" 2>&1
echo "now done"
if [ -n "$REF" ]; then
  export VRC_HIP_LIB="$PWD/$REF"
  timeout -k 10 500 bash tools/profile.sh "$OUT/ref" -- python3 tools/dev_bench.py --kernels 3 --filters 1 --steps 5
  python3 tools/prof_summary.py "$OUT/ref" vrc_k_raycast_lds "" > "$OUT/summary_ref.txt" 2>&1
  echo "ref done"
fi
