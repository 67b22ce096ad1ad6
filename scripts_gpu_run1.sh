set -x
mkdir -p gpurun_out
export TMPDIR=/tmp
rocminfo | grep -E 'Marketing|gfx|Compute Unit' | head -6 > gpurun_out/rocminfo.txt 2>&1
nproc >> gpurun_out/rocminfo.txt; lscpu | grep 'Model name' >> gpurun_out/rocminfo.txt
timeout 1500 python -m pytest tests -m gpu -q --maxfail=40 -x --deselect tests/test_gpu_fullsize.py > gpurun_out/pytest_gpu.log 2>&1; echo "pytest exit $?" >> gpurun_out/pytest_gpu.log
tail -40 gpurun_out/pytest_gpu.log
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1; tail -3 gpurun_out/smoke.log
AVIFGPU_HOT_VARIANT=1 timeout 600 python bench.py --steps 100 --warmup 10 > gpurun_out/bench_v1.json 2> gpurun_out/bench_v1.err; cat gpurun_out/bench_v1.json; tail -3 gpurun_out/bench_v1.err
AVIFGPU_HOT_VARIANT=0 timeout 600 python bench.py --steps 100 --warmup 10 --no-cpu-baseline > gpurun_out/bench_v0.json 2> gpurun_out/bench_v0.err; cat gpurun_out/bench_v0.json; tail -3 gpurun_out/bench_v0.err
timeout 600 python bench.py --steps 50 --warmup 5 --no-cpu-baseline --chroma 420 > gpurun_out/bench_420.json 2>&1; cat gpurun_out/bench_420.json
