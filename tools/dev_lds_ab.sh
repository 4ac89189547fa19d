#!/bin/bash
# Developer A/B of the LDS-staged kernel ON the GPU box: region shapes x library variants, trilinear.
# usage: tools/dev_lds_ab.sh OUT "lib1 lib2 ..." "shape1 shape2 ..." [extra dev_bench args]
# stops at the first configuration that produces no timing (a hung or failing kernel must not be run again)
OUT=$1; LIBS=$2; SHAPES=$3; shift; shift; shift
mkdir -p "$OUT"
for lib in $LIBS; do
  for shape in $SHAPES; do
    for vol in mem hash; do
      for spin in "0 0" "0.5 0.35"; do
        tag="$(basename $lib .so) shape=$shape vol=$vol spin=$spin"
        r=$(VRC_LDS_SHAPE=$shape VRC_HIP_LIB=$lib timeout -k 10 90 python tools/dev_bench.py --kernels 3 --filters 1 --steps 8 --volume $vol --spin $spin "$@" 2>&1 | grep "kernel 3" | sed 's/ -> .*//')
        echo "$tag :: $r" | tee -a "$OUT/ab.txt"
        if [ -z "$r" ]; then echo "no timing: stopping" | tee -a "$OUT/ab.txt"; exit 1; fi
      done
    done
  done
done
