mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_icc.py tests/test_gpu_write.py -m gpu -q --maxfail=10 2>&1 | tail -4
python tools/bench_configs.py 2>/dev/null | grep -E "ICC|C4 8192.2 RGB f32 -> 10-bit PQ 4:4:4" > gpurun_out/configs_icc.jsonl; cat gpurun_out/configs_icc.jsonl
