mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -q --maxfail=10 2>&1 | tail -15
python bench.py --steps 100 --warmup 20 --no-cpu-baseline 2>/dev/null | cut -c1-400
