#!/bin/bash
# Round 5, call 2: the GPU suite on the fresh-data policy defaults; every table row on fresh data with its math-free twin (tree) and its
# MEMORY-free twin (the AG_MATH_ONLY build: compute side alone); what the box gives the bare C4 pattern (tools/membench_r02); first launch of
# a process with plain and compressed (--offload-compress) code objects.  Results: gpurun_out/r05b/.
out=gpurun_out/r05b; mkdir -p $out
V=$PWD/avif-format_amd/variants
timeout 600 python -m pytest tests -m gpu -x -q > $out/pytest_gpu.txt 2>&1; tail -3 $out/pytest_gpu.txt
python tools/bench_configs.py > $out/bench_configs_tree.jsonl 2> $out/bench_configs_tree.err; wc -l $out/bench_configs_tree.jsonl
AVIFGPU_LIB=$V/libavifgpu_mathonly.so BENCH_SAME=0 BENCH_TWIN=0 BENCH_FOOTPRINT_GB=0.1 python tools/bench_configs.py > $out/bench_configs_mathonly.jsonl 2> $out/bench_configs_mathonly.err; wc -l $out/bench_configs_mathonly.jsonl
tools/membench_r02 patterns > $out/membench_patterns.txt 2>&1
tools/membench_r02 rotate > $out/membench_rotate.txt 2>&1
head -14 $out/membench_patterns.txt; grep -A3 "buffer sets: 4" $out/membench_rotate.txt | head -8
for v in tree zc tree zc; do
  lib=$V/libavifgpu_$v.so; [ $v = tree ] && lib=$PWD/avif-format_amd/libavifgpu.so
  AVIFGPU_LIB=$lib python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-pcie --no-c5 --no-live-traffic --no-pattern 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=d['cold_launch']
print('first launch $v: event %.3f ms wall %.3f ms | cold %.4f ms | frac %.4f' % (c['first_launch_in_process_ms'], c['first_launch_in_process_wall_ms'], c['cold_first_launch_ms'], d['roofline']['frac']))"
done | tee $out/first_launch_plain_vs_compressed.txt
python - <<'PY'
import json
t={json.loads(l)['config']:json.loads(l) for l in open('gpurun_out/r05b/bench_configs_tree.jsonl')}
for l in open('gpurun_out/r05b/bench_configs_mathonly.jsonl'):
    m=json.loads(l); k=t.get(m['config'])
    if k: print('%-92s kernel %.4f ms %.3f | math only %.4f ms (%.2f of kernel) | twin %s' % (m['config'][:92], k['ms_mean'], k['frac_of_8TBs'], m['ms_mean'], m['ms_mean']/k['ms_mean'], k.get('twin_ms_mean')))
PY
