timeout 900 python -m pytest tests/test_gpu_write.py tests/test_gpu_tiles.py tests/test_gpu_fuzz.py tests/test_gpu_host_shim.py tests/test_cli.py tests/test_gpu_extremes.py tests/test_abi.py -m gpu -q -x 2>&1 | tail -3
python tools/bench_configs.py "REF " "RGBA8 premultiplied" 2>/dev/null | cut -c1-80,110-330
