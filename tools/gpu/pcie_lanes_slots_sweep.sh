#!/bin/bash
# Round 3, pass q: ordered uploads with lanes x slots <= 4 streams per device (ROCm multiplexes streams onto 4 hardware queues by default)
out=gpurun_out/r03q; mkdir -p $out
for cfg in "1 2 2" "1 2 3" "1 1 4" "1 1 6" "1 4 1" "0 2 2" "1 3 2"; do set -- $cfg; for chunk in 16 32; do
AVIFGPU_UPLOAD_DEPTH=$1 AVIFGPU_LANES=$2 AVIFGPU_SLOTS=$3 timeout 120 python - 2>/dev/null <<PY
import sys, json, io, contextlib
sys.path.insert(0, "tools"); sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import bench_pcie
buf = io.StringIO()
with contextlib.redirect_stdout(buf):
    bench_pcie.run(8192, 8192, 1, $chunk, True, reps=6, slots=$3)
    bench_pcie.run(8192, 8192, 1, $chunk, False, reps=4, slots=$3)
for l in buf.getvalue().splitlines():
    d = json.loads(l); print(json.dumps({"upload_depth": $1, "lanes": $2, "slots": $3, "chunk_MiB": $chunk, "memory": d["memory"], "ms": round(d["seconds"] * 1e3, 2), "H2D_GB_s": d["H2D_GB_s"]}))
PY
done; done | tee $out/lanes_slots_sweep.jsonl
