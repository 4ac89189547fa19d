#!/bin/bash
# Developer A/B of trilinear kernel forms ON the GPU box: library variants (tools/build_variants.sh / dev_layouts.sh:
# VRC_LDS_ROWS, VRC_LDS_OCC, VRC_LDS_STAGE_N, VRC_LDS_KMAX, VRC_PGROUP, ... are COMPILE-time switches, one library
# each) x kernels x volumes x views.
# usage: tools/dev_lds_ab.sh OUT "lib1 lib2 ..." "kernel1 kernel2 ..." [extra dev_bench args]     (kernels: 3 = LDS-staged, 5 = tap-packed)
# stops at the first configuration that produces no timing (a hung or failing kernel must not be run again)
OUT=$1; LIBS=$2; KERNELS=$3; shift; shift; shift
mkdir -p "$OUT"
for lib in $LIBS; do
  for k in $KERNELS; do
    for vol in mem hash; do
      for spin in "0 0" "0.5236 0.349"; do
        tag="$(basename $lib .so) kernel=$k vol=$vol spin=$spin"
        r=$(VRC_HIP_LIB=$lib timeout -k 10 90 python tools/dev_bench.py --kernels $k --filters 1 --steps 8 --volume $vol --spin $spin "$@" 2>&1 | grep "kernel $k" | sed 's/ -> .*//')
        echo "$tag :: $r" | tee -a "$OUT/ab.txt"
        if [ -z "$r" ]; then echo "no timing: stopping" | tee -a "$OUT/ab.txt"; exit 1; fi
      done
    done
  done
done
