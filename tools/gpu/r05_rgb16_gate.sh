for rep in 1 2; do for hv in 7 15; do echo "== hv$hv (pass $rep)"; BENCH_HOT_VARIANT=$hv BENCH_TWIN=0 BENCH_SAME=0 python tools/bench_configs.py "GEO 6000x4000 RGB16" "W16 8192^2 RGB16" "C3 8192" "D12 8192^2 RGB16" "SZ16" 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    try: d=json.loads(l)
    except Exception: continue
    print('%-88s %.4f ms  %.3f  %s' % (d['config'][:88], d['ms_mean'], d['frac_of_8TBs'], d['kernel'][:38]))"; done; done
