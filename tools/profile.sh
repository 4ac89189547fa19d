#!/bin/bash
# rocprofv3 passes over a benchmark command (run ON the GPU box via gpurun).
# usage: tools/profile.sh OUTDIR -- python3 <script> [args]     (program itself after --, no wrappers)
# PMC counters are collected in their own passes, never together with --kernel-trace/--stats.
set -u
OUT=$1; shift; shift
REPO=${GRAFT_REPO_ROOT:-$PWD}
mkdir -p "$REPO/$OUT"
cd /tmp && export TMPDIR=/tmp
run() { # name, rocprof args...
  local name=$1; shift
  (cd "$REPO" && rocprofv3 "$@" --output-format csv -d "$REPO/$OUT/$name" -- "${CMD[@]}" > "$REPO/$OUT/$name.log" 2>&1)
  echo "pass $name rc=$?"
}
CMD=("$@")
run trace --kernel-trace --stats
run pmc_sq1 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAIT_ANY
run pmc_sq2 --pmc SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM_RD SQ_THREAD_CYCLES_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE
run pmc_fetch --pmc FETCH_SIZE
run pmc_write --pmc WRITE_SIZE
run pmc_tcc --pmc TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum
# what the L2s ask the fabric / HBM for, by request size: FETCH_SIZE tallies every request at 64 bytes
run pmc_ea --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum
run pmc_ta --pmc TA_TA_BUSY_sum TA_FLAT_READ_WAVEFRONTS_sum
run pmc_ta2 --pmc TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum
run pmc_tcp --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum
run pmc_tlb --pmc TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum
run pmc_grbm --pmc GRBM_GUI_ACTIVE GRBM_TA_BUSY
