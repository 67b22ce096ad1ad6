#!/bin/bash
# Wave-cycle accounting of chosen rows (waiting on data / waiting for an issue slot / LDS bank conflicts): one --pmc pass per counter pair.
#   tools/gpu/r04_pmc_reads.sh OUT.json "row substring" ...
mkdir -p gpurun_out/r04
export BENCH_TWIN=0
out=$1; shift
PMC_GROUPS="SQ_WAVE_CYCLES,SQ_BUSY_CYCLES;SQ_WAIT_INST_ANY,SQ_WAIT_ANY;SQ_ACTIVE_INST_VALU,SQ_ACTIVE_INST_LDS;SQ_INSTS_LDS,SQ_LDS_BANK_CONFLICT;SQ_WAIT_INST_LDS,SQ_LDS_IDX_ACTIVE;SQ_INSTS_VALU,SQ_INSTS_SALU" \
  python tools/gpu/pmc_rows.py "$out" "$@" > "${out%.json}.log" 2>&1
