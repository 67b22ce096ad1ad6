mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -q --maxfail=10 2>&1 | tail -3
python tools/bench_configs.py 2>/dev/null > gpurun_out/configs.jsonl; cat gpurun_out/configs.jsonl
# world_size 1 through torch.distributed.run exactly as the driver launches it
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 1 --steps 50 --warmup 5 --no-cpu-baseline 2>&1 | tail -2 | cut -c1-300
