mkdir -p gpurun_out
for h in 1024 2048 8192; do
  echo "== C4 tile height $h: hot-kernel tuning words (7 = 8 px/lane NT, 5 = 4 px/lane NT, caps in bits 8..)"
  python bench.py --height $h --steps 50 --warmup 10 --no-c5 --no-pcie --no-cpu-baseline --sweep "7,5,0x80007,0x100007,0x200007,0x400007" 2>&1 >/dev/null | grep sweep
done
for v in "" w8nc4; do
  if [ -z "$v" ]; then lib=""; else lib=$PWD/avif-format_amd/variants/libavifgpu_$v.so; fi
  echo "== variant ${v:-base}"
  AVIFGPU_LIB=$lib python tools/bench_configs.py "C2" "W8 8192^2 RGB8" "GEO 7952x5304 RGB8" "BIG 16384^2 RGB8" 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    r = json.loads(l); print('%-70s %8.4f ms  %.3f' % (r['config'][:70], r['ms_mean'], r['frac_of_8TBs']))"
done
