mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_write.py tests/test_gpu_tiles.py tests/test_gpu_fullsize.py -m gpu -q --maxfail=10 -s 2>&1 | grep -E "PQ sweep|passed|failed|FAILED|rror" | head -20
python bench.py --steps 100 --warmup 10 --no-cpu-baseline --sweep 0x7,0x5,0x17,0x3,0x400007,0x1000007 > gpurun_out/b7.json 2> gpurun_out/b7.txt; grep sweep gpurun_out/b7.txt; cat gpurun_out/b7.json
python bench.py --steps 100 --warmup 10 --no-cpu-baseline --chroma 420 > gpurun_out/b7_420.json 2>/dev/null; cat gpurun_out/b7_420.json
