timeout 600 python -m pytest tests/test_gpu_write.py tests/test_gpu_kernel_equivalence.py tests/test_gpu_fullsize.py tests/test_gpu_extremes.py -m gpu -q -x 2>&1 | tail -2
for v in "" sb64 sb256; do
  if [ -z "$v" ]; then lib=""; else lib=$PWD/avif-format_amd/variants/libavifgpu_$v.so; fi
  echo "== stream block ${v:-128(base)}"
  AVIFGPU_LIB=$lib python tools/bench_configs.py "C4 8192" "BIG 16384^2 RGB f32" "C5" "REF RGB16" "REF RGBA16" "GEO 7952x5304 RGB f32" "GEO 6000x4000 RGB f32" 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    r = json.loads(l); print('%-78s %8.4f ms  %.3f' % (r['config'][:78], r['ms_mean'], r['frac_of_8TBs']))"
done
