#!/bin/bash
# Developer A/B (run ON the GPU box): the brick upload's repack kernel under two library builds.
# usage: tools/dev_ab_upload.sh OUT lib1 lib2 ...
OUT=$1; shift
REPO=${GRAFT_REPO_ROOT:-$PWD}
mkdir -p "$REPO/$OUT"
export TMPDIR=/tmp
for lib in "$@"; do
  name=$(basename $lib .so)
  export VRC_HIP_LIB=$REPO/$lib
  (cd "$REPO" && rocprofv3 --kernel-trace --stats --output-format csv -d "$REPO/$OUT/$name" -- python3 tools/dev_upload.py 512 > "$REPO/$OUT/$name.log" 2>&1)
  echo "== $name"; python3 -c "
import csv, glob, sys
for f in glob.glob(sys.argv[1] + '/*/*kernel_stats.csv'):
    for r in csv.DictReader(open(f)):
        if 'pack' in r['Name']:
            print(r['Name'][:40], 'calls', r['Calls'], 'avg_ns', r['AverageNs'], 'min', r['MinNs'], 'max', r['MaxNs'])
" "$REPO/$OUT/$name"
  rm -rf "$REPO/$OUT/$name"
done
