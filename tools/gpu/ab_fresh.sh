#!/bin/bash
# Round 5: A/B of bench_configs rows over library builds on FRESH data (rotating buffer sets), interleaved twice on one box:
#   ONLY="pat|pat" [SAME=1] [REPS=n] tools/gpu/ab_fresh.sh name1 name2 ...     (avif-format_amd/variants/libavifgpu_<name>.so; "tree" = this tree's)
IFS="|" read -ra pats <<< "${ONLY:-C4 8192}"
export BENCH_TWIN=0 BENCH_SAME=${SAME:-0}
for rep in $(seq 1 ${REPS:-2}); do
for v in "$@"; do
  lib=$PWD/avif-format_amd/variants/libavifgpu_$v.so
  [ "$v" = tree ] && lib=$PWD/avif-format_amd/libavifgpu.so
  echo "== $v (pass $rep)"
  AVIFGPU_AB_OLD_LIB=1 AVIFGPU_LIB=$lib python tools/bench_configs.py "${pats[@]}" 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    try: d=json.loads(l)
    except Exception: continue
    print('%-88s %.4f ms  %.3f  %s%s' % (d['config'][:88], d['ms_mean'], d['frac_of_8TBs'], d['kernel'][:38], ('  | same %.3f' % d['frac_same']) if 'frac_same' in d else ''))"
done; done
