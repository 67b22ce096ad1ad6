mkdir -p gpurun_out
timeout 900 python -X faulthandler -m pytest tests/test_gpu_icc.py tests/test_icc_golden.py tests/test_icc16.py tests/test_icc8.py tests/test_gpu_write.py tests/test_cli.py -m gpu -q --maxfail=10 2>&1 | tail -6
python tools/bench_configs.py ICC "16-bit doc" "8-bit doc" W16 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    r = json.loads(l); print('%-95s %-62s %8.4f ms  %.3f' % (r['config'][:95], r['kernel'][:62], r['ms_mean'], r['frac_of_8TBs']))"
