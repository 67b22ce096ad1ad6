mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -q --maxfail=10 2>&1 | tail -3
python tools/bench_configs.py 2>/dev/null > gpurun_out/configs.jsonl; cat gpurun_out/configs.jsonl
