#!/bin/bash
# exact-match fractions of tests/test_gpu_icc.py (against the real lcms2) per library variant: mean and minimum per test family
for v in "$@"; do
  echo "== $v"
  AVIFGPU_LIB=$PWD/avif-format_amd/variants/libavifgpu_$v.so timeout 600 python -m pytest tests/test_gpu_icc.py -m gpu -q -x -s 2>&1 | grep -E "^icc->" | python -c "
import sys, collections
acc = collections.defaultdict(list)
for l in sys.stdin:
    w = l.split()
    key = w[0] + ' ' + ('linear' if 'linear' in w[1] else 'parametric')
    acc[key].append(float(w[w.index('exact') + 1]))
for k, v in sorted(acc.items()): print('%-28s n=%2d  mean exact %.5f  min %.5f' % (k, len(v), sum(v)/len(v), min(v)))"
done
