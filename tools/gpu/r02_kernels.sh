# kernel pass: parity of the changed kernels, then their timings; PCIe path after the worker polling fix
mkdir -p gpurun_out
timeout 900 python -X faulthandler -m pytest tests -m gpu -q --maxfail=10 2>&1 | tail -12
python tools/bench_configs.py ICC "16-bit doc" "8-bit doc" GEO "C4 8192" "C5" 2>/dev/null | tee gpurun_out/configs_kernels.jsonl | python -c "
import sys, json
for l in sys.stdin:
    r = json.loads(l); print('%-95s %-62s %8.4f ms  %.3f' % (r['config'][:95], r['kernel'][:62], r['ms_mean'], r['frac_of_8TBs']))"
for lanes in 1 2; do
  AVIFGPU_LANES=$lanes timeout 120 python - 2>/dev/null <<'PY'
import sys, os
sys.path.insert(0, "tools"); sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import bench_pcie
bench_pcie.run(8192, 8192, 1, 32, True)
bench_pcie.run(8192, 8192, 1, 32, False)
PY
done | tee gpurun_out/lanes2.jsonl | cut -c100-400
AVIFGPU_LANES=1 timeout 300 python tools/bench_host_shim.py floor 2>/dev/null | cut -c60-330
timeout 300 python tools/bench_host_shim.py floor 2>/dev/null | cut -c60-330
