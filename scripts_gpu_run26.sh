timeout 900 python -m pytest tests/test_gpu_fullsize_roundtrip.py -m gpu -q --maxfail=10 2>&1 | tail -15
