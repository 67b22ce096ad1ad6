#!/bin/bash
# Round 3, pass e: suite, t2 sweep per pq_evaluation, ICC rows, counters of the 16-bit ICC rows, the round's bench profile.
out=gpurun_out/r03e; mkdir -p $out
fmt='import sys,json
for l in sys.stdin:
    try: d=json.loads(l)
    except Exception: continue
    print("%-84s %.4f ms  %.3f  %s" % (d["config"][:84], d["ms_mean"], d["frac_of_8TBs"], d["kernel"][:70]))'
python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > $out/pytest.txt
python -m pytest tests/test_gpu_t2_truth.py -q -s -k "pq_write" 2>&1 | grep -E "PQ OETF|mismatches|passed|failed" > $out/t2.txt
python tools/bench_configs.py "ICC" "R16 8192^2 12-bit mono" "R8 8192^2 8-bit 4:2:0 BT.601 + alpha" "R16 8192^2 12-bit 4:4:4" 2>/dev/null | python -c "$fmt" > $out/rows.txt
python tools/gpu/pmc_rows.py $out/pmc_icc16.json "16-bit doc + ICC" > $out/pmc_icc16.log 2>&1
bash tools/gpu/profile_r03.sh > $out/profile.log 2>&1
cat $out/pytest.txt $out/t2.txt $out/rows.txt; tail -60 $out/pmc_icc16.log; tail -5 $out/profile.log
