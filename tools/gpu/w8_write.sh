#!/bin/bash
# parity of the write path + the 8-bit write rows of tools/bench_configs.py (one line each)
timeout 900 python -m pytest tests/test_gpu_write.py tests/test_gpu_kernel_equivalence.py tests/test_gpu_host_shim.py tests/test_gpu_extremes.py tests/test_gpu_fullsize.py tests/test_gpu_tiles.py -m gpu -q -x 2>&1 | tail -2
IFS="|" read -ra pats <<< "${ONLY:-C2|W8 8192|GEO 7952x5304 RGB8}"
python tools/bench_configs.py "${pats[@]}" 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    try: d=json.loads(l)
    except Exception: continue
    print('%-84s %.4f ms  %.3f' % (d['config'][:84], d['ms_mean'], d['frac_of_8TBs']))"
