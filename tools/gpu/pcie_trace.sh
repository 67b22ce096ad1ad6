# PCIe-path investigation: suite first, then where a tile's time goes (AVIFGPU_TRACE), lanes, shim floor
mkdir -p gpurun_out
timeout 900 python -X faulthandler -m pytest tests -m gpu -q --maxfail=10 2>&1 | tail -6
AVIFGPU_TRACE=1 timeout 120 python - > gpurun_out/trace_pinned_1ctx.txt 2>&1 <<'PY'
import sys, os
sys.path.insert(0, "tools"); sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import bench_pcie
bench_pcie.run(8192, 8192, 1, 32, True, reps=2)
PY
tail -30 gpurun_out/trace_pinned_1ctx.txt
for lanes in 1 2 3; do
  AVIFGPU_LANES=$lanes timeout 120 python - 2>/dev/null <<'PY'
import sys, os
sys.path.insert(0, "tools"); sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import bench_pcie
bench_pcie.run(8192, 8192, 1, 32, True)
bench_pcie.run(8192, 8192, 1, 32, False)
PY
done | tee gpurun_out/lanes.jsonl
for lanes in 1 2; do AVIFGPU_LANES=$lanes timeout 300 python tools/bench_host_shim.py floor 2>/dev/null; done | tee gpurun_out/host_shim_floor.jsonl | cut -c1-330
AVIFGPU_LANES=2 timeout 300 python tools/bench_host_shim.py write 2>/dev/null | tee gpurun_out/host_shim_write.jsonl | cut -c1-330
AVIFGPU_LANES=2 timeout 300 python tools/bench_host_shim.py read 2>/dev/null | tee gpurun_out/host_shim_read.jsonl | cut -c1-330
