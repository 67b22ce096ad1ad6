set -x
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_write.py tests/test_gpu_tiles.py -m gpu -q --maxfail=10 > gpurun_out/pytest_gpu.log 2>&1; tail -3 gpurun_out/pytest_gpu.log
timeout 900 python bench.py --steps 50 --warmup 5 --no-cpu-baseline --sweep 0x1,0x3,0x5,0x7,0x9,0xb,0xd,0xf,0x11,0x13,0x15,0x17,0x19,0x1b,0x1d,0x1f,0x100005,0x100007,0x10000d,0x10000f,0x100015,0x100017,0x10001d,0x10001f,0x1000005,0x1000007,0x100000d,0x100000f,0x1000015,0x1000017,0x100001d,0x100001f > gpurun_out/sweep_pq.json 2> gpurun_out/sweep_pq.txt; grep sweep gpurun_out/sweep_pq.txt
timeout 900 python bench.py --steps 50 --warmup 5 --no-cpu-baseline --transfer clip --sweep 0x7,0xf,0x17,0x5,0x100007,0x1000007 > gpurun_out/sweep_clip.json 2> gpurun_out/sweep_clip.txt; grep sweep gpurun_out/sweep_clip.txt
