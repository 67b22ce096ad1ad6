for rep in 1 2; do
for v in "" hb64 hb128 hb512 hb1024; do
  if [ -z "$v" ]; then lib=""; else lib=$PWD/avif-format_amd/variants/libavifgpu_$v.so; fi
  AVIFGPU_LIB=$lib python bench.py --steps 300 --warmup 30 --no-cpu-baseline --no-pcie --no-c5 2>gpurun_out/hb_err.txt | python -c "
import sys, json
s = sys.stdin.read()
try:
    r = json.loads(s); print('block ${v:-256(base)}  ms_per_step', r['ms_per_step'], 'frac', r['roofline']['frac'])
except Exception as e:
    print('variant ${v:-base} failed', s[:200]); print(open('gpurun_out/hb_err.txt').read()[-600:])"
done
done
