#!/bin/bash
# A/B of library variants (tools/ab_variants.sh) on tools/bench_configs.py rows, interleaved twice on the same box:
#   ONLY="pat|pat" tools/gpu/ab_r8.sh variantA variantB ...
IFS="|" read -ra pats <<< "${ONLY:-R8 8192}"
for rep in 1 2; do
for v in "$@"; do
  echo "== $v"
  AVIFGPU_LIB=$PWD/avif-format_amd/variants/libavifgpu_$v.so python tools/bench_configs.py "${pats[@]}" 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    try: d=json.loads(l)
    except Exception: continue
    print('%-84s %.4f ms  %.3f' % (d['config'][:84], d['ms_mean'], d['frac_of_8TBs']))"
done; done
