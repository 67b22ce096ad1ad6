#!/bin/bash
# Round 3, pass m: single-instruction issue rates (alubench) and the instruction / LDS / wait counters of the rows still under 0.70:
# u8 reads, C2 at its real size, the parametric-TRC ICC streaming kernel.
out=gpurun_out/r03m; mkdir -p $out
true
PMC_GROUPS="SQ_INSTS_VALU,SQ_INSTS_SALU,SQ_INSTS_LDS,SQ_INSTS_VMEM_RD,SQ_INSTS_VMEM_WR;SQ_LDS_BANK_CONFLICT,SQ_LDS_IDX_ACTIVE;SQ_WAVE_CYCLES,SQ_BUSY_CYCLES,SQ_WAVES;SQ_WAIT_INST_ANY,SQ_WAIT_ANY,SQ_ACTIVE_INST_ANY;SQ_INST_CYCLES_VMEM,SQ_ACTIVE_INST_VALU,SQ_ACTIVE_INST_LDS" \
  timeout 1200 python tools/gpu/pmc_rows.py $out/pmc_reads_c2_icc2.json "R8 8192^2" "GEO 7952x5304 8-bit" "C2 4096" "C4 + ICC (sRGB parametric" "R32 8192^2 10-bit mono" > $out/pmc.log 2>&1
tail -3 $out/pmc.log
python - <<PY
import json
d=json.load(open('$out/pmc_reads_c2_icc2.json'))
for k,v in d['rows'].items():
    print(k[:70]); print('   ', v.get('kernel','')[:100])
    print('   ', {a:(round(b) if isinstance(b,(int,float)) else b) for a,b in v.items() if a not in ('kernel',)})
print(d.get('notes'))
PY
