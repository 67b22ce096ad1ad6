#!/bin/bash
# Round 3, pass j: parity of the caller-supplied 16-bit table (A2B profiles), integer VALU issue rates for the ICC stage (alubench),
# and the instruction / LDS counters of the two table kernels (8-bit matrix-shaper icc=3, 16-bit CLUT icc=5).
out=gpurun_out/r03j; mkdir -p $out
timeout 900 python -m pytest tests/test_icc16.py tests/test_abi.py -m gpu -q -x 2>&1 | tail -3 > $out/pytest.txt; cat $out/pytest.txt
./tools/alubench > $out/alubench.txt 2>&1; cat $out/alubench.txt
PMC_GROUPS="SQ_INSTS_VALU,SQ_INSTS_SALU,SQ_INSTS_LDS;SQ_LDS_BANK_CONFLICT,SQ_LDS_IDX_ACTIVE,SQ_LDS_ADDR_CONFLICT;SQ_ACTIVE_INST_VALU,SQ_ACTIVE_INST_LDS,SQ_WAVE_CYCLES,SQ_BUSY_CYCLES;SQ_INSTS_VMEM_RD,SQ_INSTS_VMEM_WR,SQ_ACTIVE_INST_VMEM,SQ_WAIT_INST_LDS" \
  timeout 1200 python tools/gpu/pmc_rows.py $out/pmc_icc3_icc5.json "8-bit doc + ICC" "16-bit doc + ICC" > $out/pmc.log 2>&1
tail -5 $out/pmc.log; python -c "
import json; d=json.load(open('$out/pmc_icc3_icc5.json'))
print(json.dumps(d, indent=1)[:6000])"
