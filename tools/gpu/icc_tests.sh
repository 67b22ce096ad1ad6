timeout 900 python -m pytest tests/test_icc8.py tests/test_gpu_icc.py tests/test_cli.py -m gpu -q --maxfail=10 2>&1 | tail -12
python tools/bench_configs.py ICC 2>/dev/null
