#!/bin/bash
# Host-pointer job from PAGEABLE caller memory: copy helpers 0 / 1 / 3 / 7, both directions; parity of the scheduler tests first
out=gpurun_out/pcie_pageable; mkdir -p $out
timeout 900 python -m pytest tests/test_gpu_multidevice.py tests/test_gpu_host_shim.py -m gpu -q -x 2>&1 | tail -2 | tee $out/pytest.txt
for t in 0 1 3 7; do
AVIFGPU_COPY_THREADS=$t timeout 200 python - 2>/dev/null <<PY
import sys, json, io, contextlib, os
sys.path.insert(0, "tools"); sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import bench_pcie
os.environ.pop("AVIFGPU_CHUNK_MB", None)
buf = io.StringIO()
with contextlib.redirect_stdout(buf):
    bench_pcie.run(8192, 8192, 1, 32, False, reps=5)
    bench_pcie.run(8192, 8192, 1, 32, True, reps=5)
    bench_pcie.run_read(8192, 8192, 32, False, reps=5)
for l in buf.getvalue().splitlines():
    d = json.loads(l); print(json.dumps({"copy_threads": $t, "config": d["config"][:40], "memory": d["memory"], "ms": round(d["seconds"] * 1e3, 2)}))
PY
done | tee $out/copy_threads.jsonl
