timeout 900 python -X faulthandler -m pytest tests -m gpu -q --maxfail=10 2>&1 | tail -5
python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-pcie 2>/dev/null | python -c "
import sys, json
r = json.loads(sys.stdin.read()); print('C4', r['ms_per_step'], r['roofline']['frac'], 'C5', r['c5']['ms_per_step'], r['c5']['per_gpu_frac_of_8TBs'])"
python tools/bench_configs.py "C4 8192" "Gray32" "R32 8192^2 10-bit 4:4:4" 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    r = json.loads(l); print('%-80s %8.4f ms  %.3f' % (r['config'][:80], r['ms_mean'], r['frac_of_8TBs']))"
