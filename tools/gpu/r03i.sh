#!/bin/bash
# Round 3, pass i: are the short rows (u8 reads, C2 4096^2) measured below the steady clock?  150 vs 3000 warm-up launches, interleaved.
out=gpurun_out/r03i; mkdir -p $out
fmt='import sys,json
for l in sys.stdin:
    try: d=json.loads(l)
    except Exception: continue
    print("%-84s %.4f ms  %.3f  %s" % (d["config"][:84], d["ms_mean"], d["frac_of_8TBs"], d["kernel"][:60]))'
for rep in 1 2; do
for w in 150 3000; do
  echo "== warm $w (rep $rep)"
  BENCH_WARM=$w python tools/bench_configs.py "R8 8192" "C2" "R16 8192^2 12-bit mono" "BIG 16384^2 8-bit" "W8 8192^2 RGB8 -> 8-bit 4:2:2" 2>/dev/null | python -c "$fmt"
done; done > $out/warm_ab.txt 2>&1
cat $out/warm_ab.txt
