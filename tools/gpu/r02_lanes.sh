timeout 300 python -m pytest tests/test_gpu_multidevice.py -m gpu -q 2>&1 | tail -2
for lanes in 2 3 4; do
  AVIFGPU_LANES=$lanes timeout 120 python - 2>/dev/null <<'PY'
import sys, os
sys.path.insert(0, "tools"); sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import bench_pcie
for chunk in (8, 16):
    bench_pcie.run(8192, 8192, 1, chunk, True, reps=6)
bench_pcie.run(8192, 8192, 1, 16, False, reps=6)
PY
done | python -c "
import sys, json
for i, l in enumerate(sys.stdin):
    r = json.loads(l); print('lanes', 2 + i // 3, r['memory'], 'chunk', r['chunk_MiB'], 'ms', r['seconds'] * 1e3, 'H2D GB/s', r['H2D_GB_s'])"
