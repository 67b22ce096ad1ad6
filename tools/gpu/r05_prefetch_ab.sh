#!/bin/bash
# Round 5: the software-pipelined span loop of write_rgb32_ycbcr444_hot (-DAG_HOT_PREFETCH=1, variant "pf") against the tree on fresh data,
# over grid caps (BENCH_HOT_VARIANT bits 8..: blocks of 4 waves; default = one span per wave up to 131072 blocks).  Two interleaved passes.
out=gpurun_out/r05d; mkdir -p $out
V=$PWD/avif-format_amd/variants
for rep in 1 2; do for v in tree pf; do for cap in 0 1792 2048 3584 4096 8192 16384; do
  lib=$V/libavifgpu_$v.so; [ $v = tree ] && lib=$PWD/avif-format_amd/libavifgpu.so
  AVIFGPU_LIB=$lib BENCH_TWIN=0 BENCH_SAME=0 BENCH_HOT_VARIANT=$((7 + cap * 256)) python tools/bench_configs.py "C4 8192^2 RGB f32 -> 10-bit PQ 4:4:4" "BIG 16384^2 RGB f32 -> 10-bit PQ 4:4:4" "GEO 7952x5304 RGB f32 -> 10-bit PQ 4:4:4" "D12 8192^2 RGB f32 -> 12-bit PQ 4:4:4" 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    try: d=json.loads(l)
    except Exception: continue
    print('$v cap %6d pass $rep  %-52s %.4f ms  %.3f' % ($cap, d['config'][:52], d['ms_mean'], d['frac_of_8TBs']))"
done; done; done > $out/prefetch_ab.txt 2>&1
cat $out/prefetch_ab.txt
