#!/bin/bash
# A/B of read-kernel variants on the 8-bit read rows, interleaved twice on the same box
for rep in 1 2; do
for v in "$@"; do
  echo "== $v"
  AVIFGPU_LIB=$PWD/avif-format_amd/variants/libavifgpu_$v.so IFS="|" read -ra pats <<< "${ONLY:-R8 8192}"; python tools/bench_configs.py "${pats[@]}" 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    try: d=json.loads(l)
    except Exception: continue
    print('%-84s %.4f ms  %.3f' % (d['config'][:84], d['ms_mean'], d['frac_of_8TBs']))"
done; done
