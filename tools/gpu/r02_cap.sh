for rep in 1 2; do
for word in 7 0x400007 0x800007 0x1000007 0x2000007 5 3; do
  AVIFGPU_HOT_VARIANT=$word python bench.py --steps 300 --warmup 30 --no-cpu-baseline --no-pcie --no-c5 2>/dev/null | python -c "
import sys, json
r = json.loads(sys.stdin.read()); print('word $word  C4 ms', r['ms_per_step'], 'frac', r['roofline']['frac'], r['config']['kernel'])"
done
done
