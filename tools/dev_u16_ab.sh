#!/bin/bash
# Developer A/B of the 16-bit LDS-staged trilinear kernel ON the GPU box: gather form (kernel 2) against the staged form
# (kernel 3) for every library given.  usage: tools/dev_u16_ab.sh "lib1 lib2 ..."
mkdir -p gpurun_out/r3
for lib in $1; do
  for vol in mem hash; do
    for spin in "0 0" "0.5 0.35"; do
      r=$(VRC_HIP_LIB=$lib timeout -k 10 120 python tools/dev_bench.py --kernels 2 3 --filters 1 --steps 8 --dtype u16 --volume $vol --spin $spin 2>&1 | grep "kernel [23]" | sed 's/ -> .*//' | tr '\n' '|')
      echo "$(basename $lib) vol=$vol spin=$spin :: $r" | tee -a gpurun_out/r3/u16_ab.txt
      if [ -z "$r" ]; then echo "no timing: stopping"; exit 1; fi
    done
  done
done
