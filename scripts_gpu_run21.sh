mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_icc.py -m gpu -q --maxfail=10 2>&1 | tail -25
