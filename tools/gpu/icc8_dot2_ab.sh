#!/bin/bash
# 8-bit ICC packed kernel: v_dot2 form of the matrix (in-tree) against three mads (variant nodot2); parity first (all 2^24 triples, both forms)
out=gpurun_out/icc8_dot2; mkdir -p $out
timeout 900 python -m pytest tests/test_icc8.py tests/test_icc_golden.py -m gpu -q -x 2>&1 | tail -2 | tee $out/pytest.txt
for rep in 1 2; do for v in "" nodot2; do
  echo "== ${v:-in-tree}"
  if [ -n "$v" ]; then export AVIFGPU_LIB=$PWD/avif-format_amd/variants/libavifgpu_$v.so; else unset AVIFGPU_LIB; fi
  python tools/bench_configs.py "8-bit doc + ICC" 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    try: d=json.loads(l)
    except Exception: continue
    print('%-84s %.4f ms  %.3f' % (d['config'][:84], d['ms_mean'], d['frac_of_8TBs']))"
done; done | tee $out/ab.txt
