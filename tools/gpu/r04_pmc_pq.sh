#!/bin/bash
# Round 4: issue-side counters of the C4 4:4:4 / 4:2:0 kernels under three library builds (own --pmc passes, no trace domains).
out=gpurun_out/r04; mkdir -p $out
export PMC_GROUPS="SQ_INSTS_VALU,SQ_INSTS_VALU_TRANS_F32,SQ_INSTS_LDS,SQ_INSTS_SALU;SQ_ACTIVE_INST_VALU,SQ_ACTIVE_INST_LDS,SQ_ACTIVE_INST_ANY,SQ_WAVE_CYCLES;SQ_WAIT_ANY,SQ_WAIT_INST_ANY,SQ_WAIT_INST_LDS,SQ_BUSY_CYCLES;SQ_LDS_BANK_CONFLICT,SQ_LDS_IDX_ACTIVE,SQ_LDS_ADDR_CONFLICT,SQ_THREAD_CYCLES_VALU"
for v in "$@"; do
  lib=$PWD/avif-format_amd/variants/libavifgpu_$v.so
  [ "$v" = tree ] && lib=$PWD/avif-format_amd/libavifgpu.so
  AVIFGPU_LIB=$lib python tools/gpu/pmc_rows.py $out/pmc_pq_$v.json "C4 8192^2 RGB f32 -> 10-bit PQ 4:4:4" "C4 8192^2 RGB f32 -> 10-bit PQ 4:2:0" > /dev/null 2>$out/pmc_pq_$v.err
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r04/pmc_pq_*.json')):
    d=json.load(open(f))
    print(f)
    for k,v in d['rows'].items():
        print('  ',k[:50], {a:(round(b) if isinstance(b,float) else b) for a,b in v.items() if a.startswith('SQ_')})
    print('  notes', d['notes'])
PY
