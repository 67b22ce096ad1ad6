#!/bin/bash
# parity of the read path + the 8-bit read rows of tools/bench_configs.py (one line each)
timeout 600 python -m pytest tests/test_gpu_read.py tests/test_gpu_kernel_equivalence.py tests/test_gpu_host_shim.py tests/test_gpu_extremes.py -m gpu -q -x 2>&1 | tail -2
python tools/bench_configs.py --only "${1:-R8 8192}" 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    try: d=json.loads(l)
    except Exception: continue
    print('%-84s %.4f ms  %.3f' % (d['config'][:84], d['ms_mean'], d['frac_of_8TBs']))"
