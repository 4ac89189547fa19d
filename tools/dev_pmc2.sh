#!/bin/bash
# Developer probe (run ON the GPU box): PMC counter groups of the raycast kernels, one rocprofv3 pass per group (never
# together with a trace).  usage: tools/dev_pmc2.sh OUTDIR "GROUP1 counters" "GROUP2 counters" ... -- python3 tools/dev_bench.py ...
OUT=$1; shift
GROUPS_=()
while [ "$1" != "--" ]; do GROUPS_+=("$1"); shift; done
shift
REPO=${GRAFT_REPO_ROOT:-$PWD}
mkdir -p "$REPO/$OUT"
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "${GROUPS_[@]}"; do
  i=$((i+1))
  (cd "$REPO" && timeout -k 10 200 rocprofv3 --pmc $grp --output-format csv -d "$REPO/$OUT/g$i" -- "$@" > "$REPO/$OUT/g$i.log" 2>&1)
  echo "pass $i ($grp) rc=$?"
done
python3 - "$REPO/$OUT" <<'PY'
import csv, glob, sys, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + "/g*/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "vrc_k_raycast" in r["Kernel_Name"]:
            agg[r["Kernel_Name"][:70]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for kern, cs in sorted(agg.items()):
    print(kern)
    for k, v in sorted(cs.items()):
        v = v[2:] if len(v) > 4 else v
        print("   %-40s n=%d mean=%.6g" % (k, len(v), sum(v) / len(v)))
PY
