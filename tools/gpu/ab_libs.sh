#!/bin/bash
# A/B of bench_configs rows over library builds (avif-format_amd/variants/libavifgpu_<name>.so; "tree" = this tree's), interleaved twice on one box:
#   ONLY="pat|pat" tools/gpu/ab_libs.sh name1 name2 ...
IFS="|" read -ra pats <<< "${ONLY:-C4 8192}"
for rep in 1 2; do
for v in "$@"; do
  lib=$PWD/avif-format_amd/variants/libavifgpu_$v.so
  [ "$v" = tree ] && lib=$PWD/avif-format_amd/libavifgpu.so
  echo "== $v (pass $rep)"
  AVIFGPU_AB_OLD_LIB=1 AVIFGPU_LIB=$lib python tools/bench_configs.py "${pats[@]}" 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    try: d=json.loads(l)
    except Exception: continue
    print('%-84s %.4f ms  %.3f  %s' % (d['config'][:84], d['ms_mean'], d['frac_of_8TBs'], d['kernel'][:40]))"
done; done
