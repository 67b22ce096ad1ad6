#!/bin/bash
# accuracy (tests/test_gpu_icc.py against the real lcms2) and speed of library variants on the parametric ICC rows
for v in "$@"; do
  echo "== $v"
  AVIFGPU_LIB=$PWD/avif-format_amd/variants/libavifgpu_$v.so timeout 600 python -m pytest tests/test_gpu_icc.py -m gpu -q -x -s 2>&1 | grep -E "passed|failed|exact|Error|assert" | tail -6
  AVIFGPU_LIB=$PWD/avif-format_amd/variants/libavifgpu_$v.so python tools/bench_configs.py "C4 + ICC" "SDR save of a 32-bit doc" 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    try: d=json.loads(l)
    except Exception: continue
    print('%-84s %.4f ms  %.3f' % (d['config'][:84], d['ms_mean'], d['frac_of_8TBs']))"
done
