mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -q --maxfail=10 2>&1 | tail -12
python bench.py --steps 100 --warmup 20 --no-cpu-baseline | cut -c1-220
