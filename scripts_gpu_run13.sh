mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -q --maxfail=10 2>&1 | tail -3
python tools/bench_configs.py > gpurun_out/configs.jsonl 2> gpurun_out/configs.err; cat gpurun_out/configs.jsonl; tail -3 gpurun_out/configs.err
