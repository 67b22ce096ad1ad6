#!/bin/bash
# Round 3, pass p: ordered uploads (AVIFGPU_UPLOAD_DEPTH) -- the tests that drive the scheduler, then the C4 host-pointer job for depth x tile size x lanes
out=gpurun_out/r03p; mkdir -p $out
timeout 900 python -m pytest tests -m gpu -q -x -k "host or shim or multidevice or pipeline or staging or cli or topology" 2>&1 | tail -3 | tee $out/pytest.txt
for depth in 0 1; do for lanes in 1 2; do for chunk in 8 16 32 64; do
AVIFGPU_UPLOAD_DEPTH=$depth AVIFGPU_LANES=$lanes timeout 120 python - 2>/dev/null <<PY
import sys, json, io, contextlib
sys.path.insert(0, "tools"); sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import bench_pcie
buf = io.StringIO()
with contextlib.redirect_stdout(buf):
    bench_pcie.run(8192, 8192, 1, $chunk, True, reps=6)
    bench_pcie.run(8192, 8192, 1, $chunk, False, reps=4)
for l in buf.getvalue().splitlines():
    d = json.loads(l); print(json.dumps({"upload_depth": $depth, "lanes": $lanes, "chunk_MiB": $chunk, "memory": d["memory"], "ms": round(d["seconds"] * 1e3, 2), "H2D_GB_s": d["H2D_GB_s"]}))
PY
done; done; done | tee $out/upload_depth_sweep.jsonl
