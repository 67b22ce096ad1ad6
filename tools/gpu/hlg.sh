timeout 600 python -m pytest tests/test_gpu_read.py tests/test_gpu_write.py -m gpu -q -x -k "hlg or HLG or t1 or tr1" 2>&1 | tail -2
timeout 600 python -m pytest tests/test_gpu_read.py tests/test_gpu_write.py -m gpu -q -x 2>&1 | tail -2
python tools/bench_configs.py HLG 2>/dev/null | cut -c1-70,220-330
