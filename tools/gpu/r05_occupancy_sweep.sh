#!/bin/bash
# (round 6: AVIFGPU_DEBUG_LDS_PAD is compiled in only with -DAG_MEASURE=1: build the library with tools/ab_variants.sh write_kernels_p1 "-DAG_MEASURE=1" measure first)
# Round 5: how the headline kernel's fresh-data rate depends on the waves per SIMD (unused dynamic LDS limits the resident workgroups:
# 16 KiB static per 4-wave workgroup, 160 KiB per CU): pad 0 -> 8 waves per SIMD, 7000 -> 7, 11000 -> 6, 16500 -> 5, 24500 -> 4, 37500 -> 3
for rep in 1 2; do for pad in 0 7000 11000 16500 24500 37500; do
  AVIFGPU_DEBUG_LDS_PAD=$pad BENCH_TWIN=0 BENCH_SAME=0 python tools/bench_configs.py "C4 8192^2 RGB f32 -> 10-bit PQ 4:4:4" "BIG 16384^2 RGB f32 -> 10-bit PQ 4:4:4" "GEO 7952x5304 RGB f32 -> 10-bit PQ 4:4:4" 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    try: d=json.loads(l)
    except Exception: continue
    print('pad %6d pass $rep  %-48s %.4f ms  %.3f' % ($pad, d['config'][:48], d['ms_mean'], d['frac_of_8TBs']))"
done; done
