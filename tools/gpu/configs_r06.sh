#!/bin/bash
# Round 6: the per-row table of the round on FRESH data -- plain pass (rotating buffer sets + the one-set loop beside it + the read rows'
# math-free twins), the MEMORY-free twin of every row (the AG_MATH_ONLY build: tools/ab_variants.sh write_kernels,read_kernels "-DAG_MATH_ONLY=1" mathonly),
# then kernel trace + the two PMC traffic passes (tools/gpu/profile_configs.sh).  Results: gpurun_out/r06f/.
out=gpurun_out/r06f; mkdir -p $out
python tools/bench_configs.py > $out/bench_configs.jsonl 2> $out/bench_configs.err
AVIFGPU_LIB=$PWD/avif-format_amd/variants/libavifgpu_mathonly.so BENCH_SAME=0 BENCH_TWIN=0 BENCH_FOOTPRINT_GB=0.1 python tools/bench_configs.py > $out/bench_configs_mathonly.jsonl 2> $out/bench_configs_mathonly.err
PMC_COUNTERS="FETCH_SIZE WRITE_SIZE" bash tools/gpu/profile_configs.sh > $out/profile_configs.log 2>&1
cp gpurun_out/prof_cfg/configs_traffic.json gpurun_out/prof_cfg/configs_under_trace.jsonl $out/ 2>/dev/null
wc -l $out/bench_configs.jsonl $out/bench_configs_mathonly.jsonl; tail -n 95 $out/profile_configs.log | cut -c1-170
