#!/bin/bash
# bench.py launched the way the driver launches N = 2, on a 1-GPU box (both ranks share the device): the line must come out, scaling "strong"
out=gpurun_out/rehearsal; mkdir -p $out
AVIFGPU_BENCH_SHARE_DEVICE=1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29531 bench.py --gpus 2 --steps 50 --warmup 10 > $out/bench_2rank.json 2> $out/bench_2rank.err
echo rc=$?; cut -c1-1200 $out/bench_2rank.json; tail -3 $out/bench_2rank.err
