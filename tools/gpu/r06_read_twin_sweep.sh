#!/bin/bash
# Round 6, item 6: the 8-bit opens and their math-free twins over launch shapes (library variants built with tools/ab_variants.sh read_kernels_p8 ...),
# fresh data, interleaved REPS times on one box:  tools/gpu/r06_read_twin_sweep.sh tree rb128 rb64 ...
export BENCH_SAME=0 BENCH_TWIN=1
for rep in $(seq 1 ${REPS:-2}); do
for v in "$@"; do
  lib=$PWD/avif-format_amd/variants/libavifgpu_$v.so
  [ "$v" = tree ] && lib=$PWD/avif-format_amd/libavifgpu.so
  echo "== $v (pass $rep)"
  AVIFGPU_AB_OLD_LIB=1 AVIFGPU_LIB=$lib python tools/bench_configs.py "R8 8192" 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    try: d=json.loads(l)
    except Exception: continue
    print('%-76s %.4f ms  %.3f | twin %s' % (d['config'][:76], d['ms_mean'], d['frac_of_8TBs'], ('%.4f ms %.3f' % (d['twin_ms_mean'], d['twin_frac_of_8TBs'])) if 'twin_ms_mean' in d else '-'))"
done; done
