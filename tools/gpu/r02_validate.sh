mkdir -p gpurun_out
for i in 1 2 3 4 5 6; do
  timeout 600 python -X faulthandler -m pytest tests -m gpu -q -x -p no:cacheprovider > gpurun_out/stress_$i.log 2>&1
  rc=$?
  echo "suite run $i rc=$rc $(tail -1 gpurun_out/stress_$i.log | cut -c1-80)"
  if [ $rc -ne 0 ]; then grep -n "Fatal Python error\|File \"\|Current thread\|Thread 0x\|Error\|FAILED" gpurun_out/stress_$i.log | head -40; fi
done
echo "--- N=2 rehearsal (ranks share the one GPU, gloo): exercises the N>1 code path only"
AVIFGPU_BENCH_SHARE_DEVICE=1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 50 --warmup 5 2>&1 | tail -3 | cut -c1-1500
echo "--- weak scaling flag"
AVIFGPU_BENCH_SHARE_DEVICE=1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29512 bench.py --gpus 2 --steps 20 --warmup 5 --scaling weak --no-c5 2>&1 | tail -1 | cut -c1-600
echo "--- bench"
python bench.py --steps 200 --warmup 20 --no-cpu-baseline 2>/dev/null | cut -c1-1200
