timeout 900 python -m pytest tests/test_gpu_host_shim.py tests/test_cli.py tests/test_gpu_icc.py tests/test_icc8.py tests/test_icc16.py tests/test_gpu_tiles.py -m gpu -q -x 2>&1 | tail -2
python tools/bench_host_shim.py 2>/dev/null > gpurun_out/host_shim.jsonl; cat gpurun_out/host_shim.jsonl | cut -c1-200
