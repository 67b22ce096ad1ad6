#!/bin/bash
# Round 4: the per-row table of the round -- plain pass (with the read rows' math-free twins), then kernel trace + the PMC passes
# (HBM traffic and vector-ALU issue; tools/gpu/profile_configs.sh).  Results: gpurun_out/r04f/.
out=gpurun_out/r04f; mkdir -p $out
python tools/bench_configs.py > $out/bench_configs.jsonl 2> $out/bench_configs.err
bash tools/gpu/profile_configs.sh > $out/profile_configs.log 2>&1
cp gpurun_out/prof_cfg/configs_traffic.json gpurun_out/prof_cfg/configs_under_trace.jsonl $out/ 2>/dev/null
wc -l $out/bench_configs.jsonl; tail -90 $out/profile_configs.log | cut -c1-170
