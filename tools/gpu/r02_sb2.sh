for rep in 1 2 3; do
for v in "" sb64 sb256; do
  if [ -z "$v" ]; then lib=""; else lib=$PWD/avif-format_amd/variants/libavifgpu_$v.so; fi
  AVIFGPU_LIB=$lib python bench.py --steps 300 --warmup 30 --no-cpu-baseline --no-pcie 2>/dev/null | python -c "
import sys, json
r = json.loads(sys.stdin.read()); print('block ${v:-128(base)}  C4 ms', r['ms_per_step'], 'frac', r['roofline']['frac'], ' C5 ms', r['c5']['ms_per_step'], r['c5']['per_gpu_frac_of_8TBs'])"
done
done
