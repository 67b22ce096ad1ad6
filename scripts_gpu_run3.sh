set -x
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1800 python -m pytest tests -m gpu -q --maxfail=25 -s > gpurun_out/pytest_gpu.log 2>&1; echo "pytest exit $?" >> gpurun_out/pytest_gpu.log
grep -E "PQ sweep|EOTF tc|passed|failed|FAILED|Error" gpurun_out/pytest_gpu.log | head -60
./tools/membench > gpurun_out/membench.txt 2>&1; cat gpurun_out/membench.txt
B="python bench.py --steps 100 --warmup 10 --no-cpu-baseline"
AVIFGPU_HOT_VARIANT=1 timeout 600 $B > gpurun_out/bench_v1.json 2> gpurun_out/bench_v1.err; cat gpurun_out/bench_v1.json
AVIFGPU_HOT_VARIANT=0 timeout 600 $B > gpurun_out/bench_v0.json 2> gpurun_out/bench_v0.err; cat gpurun_out/bench_v0.json
