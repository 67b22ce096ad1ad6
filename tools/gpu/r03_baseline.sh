#!/bin/bash
# Round-3 opening pass on one box: GPU suite, PQ variant accuracy table, the plain bench line, every bench_configs row.
mkdir -p gpurun_out/r03a
python -m pytest tests -m gpu -x -q > gpurun_out/r03a/pytest.txt 2>&1; echo "pytest rc=$?" >> gpurun_out/r03a/pytest.txt
tail -3 gpurun_out/r03a/pytest.txt
timeout 300 tools/pq_variants > gpurun_out/r03a/pq_variants.txt 2>&1; cat gpurun_out/r03a/pq_variants.txt
python bench.py > gpurun_out/r03a/bench.json 2> gpurun_out/r03a/bench.err; cat gpurun_out/r03a/bench.json
python tools/bench_configs.py > gpurun_out/r03a/bench_configs.jsonl 2> gpurun_out/r03a/bench_configs.err
wc -l gpurun_out/r03a/bench_configs.jsonl
