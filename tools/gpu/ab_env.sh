#!/bin/bash
# A/B of bench_configs rows under two settings of an environment variable, interleaved twice on one box:
#   ONLY="pat|pat" tools/gpu/ab_env.sh VAR valueA valueB
var=$1; shift
IFS="|" read -ra pats <<< "${ONLY:-C3}"
for rep in 1 2; do
for v in "$@"; do
  echo "== $var=$v"
  env $var=$v python tools/bench_configs.py "${pats[@]}" 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    try: d=json.loads(l)
    except Exception: continue
    print('%-84s %.4f ms  %.3f  %s' % (d['config'][:84], d['ms_mean'], d['frac_of_8TBs'], d['kernel'][:40]))"
done; done
