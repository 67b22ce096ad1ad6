#!/bin/bash
# Round 4: the PQ rows, round 3's library (variants/libavifgpu_r03.so, built from the round-3 commit) against this tree's, interleaved twice on one box.
# The 12-bit rows are the plug-in's default depth (AvifFormat.cpp:95); AUTO = the close evaluation.
out=${1:-gpurun_out/r04/pq_table_form_ab.txt}
mkdir -p $(dirname $out)
pats=("C4 8192^2 RGB f32 -> 10-bit PQ" "D12 " "C5 16384" "W32 8192")
fmt='import sys,json
for l in sys.stdin:
    try: d=json.loads(l)
    except Exception: continue
    print("%-92s %.4f ms  %.3f  %s" % (d["config"][:92], d["ms_mean"], d["frac_of_8TBs"], d["kernel"][:44]))'
for rep in 1 2; do
  for lib in avif-format_amd/variants/libavifgpu_r03.so avif-format_amd/libavifgpu.so; do
    echo "== $lib (pass $rep)"
    AVIFGPU_LIB=$PWD/$lib python tools/bench_configs.py "${pats[@]}" 2>/dev/null | python -c "$fmt"
  done
done > $out 2>&1
cat $out
