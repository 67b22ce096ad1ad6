for rep in 1 2; do for hv in 7 23 5; do echo "== hv$hv (pass $rep)"; BENCH_HOT_VARIANT=$hv BENCH_TWIN=0 BENCH_SAME=0 python tools/bench_configs.py "C4 8192^2 RGB f32 -> 10-bit PQ 4:4:4" "GEO 7952x5304 RGB f32 -> 10-bit PQ 4:4:4" "GEO 6000x4000 RGB f32" "D12 8192^2 RGB f32 -> 12-bit PQ 4:2:2" "C3 8192" "GEO 6000x4000 RGB16" "R16 8192^2 10-bit 4:4:4" "R32 8192^2 10-bit 4:4:4" "GEO 6000x4000 10-bit" "D12 8192^2 12-bit 4:2:2" "TILE" 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    try: d=json.loads(l)
    except Exception: continue
    print('%-88s %.4f ms  %.3f  %s' % (d['config'][:88], d['ms_mean'], d['frac_of_8TBs'], d['kernel'][-24:]))"; done; done
