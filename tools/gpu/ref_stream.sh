timeout 900 python -m pytest tests/test_gpu_write.py tests/test_gpu_tiles.py tests/test_gpu_fuzz.py tests/test_gpu_host_shim.py tests/test_cli.py tests/test_gpu_icc.py -m gpu -q -x 2>&1 | tail -3
python tools/bench_configs.py "interleaved RRGGBB" "C4 8192^2 RGB f32 -> 10-bit PQ 4:4:4" 2>/dev/null | cut -c1-330
