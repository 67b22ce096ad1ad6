#!/bin/bash
# 16-bit table kernel: node-pair tables (in-tree) against the cell records (variant cellrec) -- parity of both against lcms2, speed, L2 counters
out=gpurun_out/icc16_layout; mkdir -p $out
bash tools/gpu/icc16_ab.sh cellrec > $out/ab.txt 2>&1; cat $out/ab.txt
AVIFGPU_LIB=$PWD/avif-format_amd/variants/libavifgpu_cellrec.so timeout 600 python -m pytest tests/test_icc16.py tests/test_icc_golden.py -m gpu -q -x 2>&1 | tail -1
PMC_GROUPS="TCC_HIT_sum,TCC_MISS_sum;FETCH_SIZE;TCP_TCC_READ_REQ_sum;SQ_INSTS_VALU,SQ_INSTS_VMEM_RD" timeout 900 python tools/gpu/pmc_rows.py $out/pmc_pair_tables.json "16-bit doc + ICC" > $out/pmc.log 2>&1
python - <<PY
import json
d=json.load(open('$out/pmc_pair_tables.json'))
for k,v in d['rows'].items():
    print(k[:60], {a:round(b) for a,b in v.items() if isinstance(b,(int,float))})
PY
