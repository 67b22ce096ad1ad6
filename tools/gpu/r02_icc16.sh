timeout 600 python -m pytest tests/test_icc16.py tests/test_icc_golden.py tests/test_gpu_icc.py -m gpu -q 2>&1 | tail -3
python tools/bench_configs.py "16-bit doc" 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    r = json.loads(l); print('%-95s %8.4f ms  %.3f' % (r['config'][:95], r['ms_mean'], r['frac_of_8TBs']))"
