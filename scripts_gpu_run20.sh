mkdir -p gpurun_out
python tools/bench_host_shim.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/host_shim.jsonl
