#!/bin/bash
# 16-bit ICC table kernel: parity (tests against the real lcms2 vectors) of the in-tree library, then speed of it and of variants
timeout 900 python -m pytest tests/test_icc16.py tests/test_gpu_icc.py tests/test_icc_golden.py -m gpu -q -x 2>&1 | tail -3
for rep in 1 2; do
for v in "" "$@"; do
  echo "== ${v:-in-tree}"
  if [ -n "$v" ]; then export AVIFGPU_LIB=$PWD/avif-format_amd/variants/libavifgpu_$v.so; else unset AVIFGPU_LIB; fi
  python tools/bench_configs.py "16-bit doc" 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    try: d=json.loads(l)
    except Exception: continue
    print('%-84s %.4f ms  %.3f' % (d['config'][:84], d['ms_mean'], d['frac_of_8TBs']))"
done; done
