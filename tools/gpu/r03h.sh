#!/bin/bash
# Round 3, pass h: suite on the tree with flat 4:2:2 launches and the sampled-curve ICC variant; rows that changed.
out=gpurun_out/r03h; mkdir -p $out
fmt='import sys,json
for l in sys.stdin:
    try: d=json.loads(l)
    except Exception: continue
    print("%-84s %.4f ms  %.3f  %s" % (d["config"][:84], d["ms_mean"], d["frac_of_8TBs"], d["kernel"][:90]))'
python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > $out/pytest.txt
python -m pytest tests/test_gpu_icc.py -q -s -k "sampled" 2>&1 | grep -E "icc-sampled|passed|failed" > $out/icc_sampled_accuracy.txt
python tools/bench_configs.py "GEO" "4:2:2" 2>/dev/null | python -c "$fmt" > $out/rows.txt
cat $out/pytest.txt $out/icc_sampled_accuracy.txt $out/rows.txt
