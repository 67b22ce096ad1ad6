timeout 600 python -m pytest tests/test_icc8.py -m gpu -q 2>&1 | tail -2
python tools/bench_configs.py "8-bit doc" 2>/dev/null | cut -c1-60,200-330
