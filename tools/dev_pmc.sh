#!/bin/bash
# Developer probe (run ON the GPU box): a few PMC counters of the raycast kernel for one libvrc_hip.so build.
# usage: tools/dev_pmc.sh OUTDIR LIB [dev_bench args...]   -- counters in separate passes, never with a trace
OUT=$1; LIB=$2; shift; shift
REPO=${GRAFT_REPO_ROOT:-$PWD}
mkdir -p "$REPO/$OUT"
cd /tmp && export TMPDIR=/tmp
export VRC_HIP_LIB="$REPO/$LIB"
for grp in "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum" "SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU" "TA_TA_BUSY_sum TA_FLAT_READ_WAVEFRONTS_sum" "GRBM_GUI_ACTIVE"; do
  name=$(echo $grp | cut -d' ' -f1)
  echo "pass $name"
  (cd "$REPO" && timeout -k 10 150 rocprofv3 --pmc $grp --output-format csv -d "$REPO/$OUT/$name" -- python3 tools/dev_bench.py --steps 5 "$@" > "$REPO/$OUT/$name.log" 2>&1)
done
python3 - "$REPO/$OUT" <<'PY'
import csv, glob, sys, collections
agg = collections.defaultdict(list)
for f in glob.glob(sys.argv[1] + "/*/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "vrc_k_raycast" in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in sorted(agg.items()):
    v = v[2:] if len(v) > 4 else v
    print("%-36s n=%d mean=%.5g" % (k, len(v), sum(v) / len(v)))
PY
