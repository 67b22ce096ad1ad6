#!/bin/bash
# Round 3, pass f: the per-row table of the round -- plain pass, then kernel trace + the two PMC passes (profile_configs.sh).
out=gpurun_out/r03f; mkdir -p $out
python tools/bench_configs.py > $out/bench_configs.jsonl 2> $out/bench_configs.err
bash tools/gpu/profile_configs.sh > $out/profile_configs.log 2>&1
cp gpurun_out/prof_cfg/configs_traffic.json gpurun_out/prof_cfg/configs_under_trace.jsonl $out/ 2>/dev/null
wc -l $out/bench_configs.jsonl; tail -75 $out/profile_configs.log | cut -c1-150
