for h in 1024 2048 4096 8192; do
  python bench.py --height $h --steps 400 --warmup 50 --no-cpu-baseline --no-pcie --no-c5 2>/dev/null | python -c "
import sys, json
r = json.loads(sys.stdin.read()); print('tile height $h  ms', r['ms_per_step'], 'frac', r['roofline']['frac'])"
done
