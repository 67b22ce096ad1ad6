#!/bin/bash
# Round 3, pass r: new scheduler defaults (2 lanes x 2 slots, uploads in order, 16 MiB sub-tiles): whole GPU suite, the host-pointer job,
# the FormatRecord shim's floor / write / read figures for 8 and 16 MiB protocol tiles, old behaviour beside them, and a copy trace.
out=gpurun_out/r03r; mkdir -p $out
timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | tail -3 | tee $out/pytest.txt
run_pcie() { timeout 120 python - 2>/dev/null <<PY
import sys
sys.path.insert(0, "tools"); sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import bench_pcie, os
c = int(os.environ.get("AVIFGPU_CHUNK_MB", "16"))
bench_pcie.run(8192, 8192, 1, c, True, reps=6)
bench_pcie.run(8192, 8192, 1, c, False, reps=4)
PY
}
echo "== defaults"; run_pcie | tee $out/pcie_defaults.jsonl
echo "== round-2 behaviour (no upload order, 2 x 4 slots, 8 MiB)"; AVIFGPU_UPLOAD_DEPTH=0 AVIFGPU_SLOTS=4 AVIFGPU_CHUNK_MB=8 run_pcie | tee $out/pcie_r02_behaviour.jsonl
for tile in 8 16; do
  echo "== shim, AVIFGPU_TILE_MB=$tile"
  AVIFGPU_TILE_MB=$tile timeout 300 python tools/bench_host_shim.py floor 2>/dev/null | cut -c1-400
  AVIFGPU_TILE_MB=$tile timeout 300 python tools/bench_host_shim.py write 2>/dev/null | cut -c1-400
  AVIFGPU_TILE_MB=$tile timeout 300 python tools/bench_host_shim.py read 2>/dev/null | cut -c1-400
done | tee $out/host_shim.txt
echo "== shim, round-2 behaviour"; AVIFGPU_UPLOAD_DEPTH=0 AVIFGPU_SLOTS=4 timeout 300 python tools/bench_host_shim.py floor 2>/dev/null | cut -c1-400 | tee $out/host_shim_r02_behaviour.txt
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; cd /tmp
timeout 300 rocprofv3 --memory-copy-trace -d $R/$out/mc -o mc --output-format csv -- bash -c "cd $R && python -c \"
import sys
sys.path.insert(0, 'tools'); sys.path.insert(0, 'tests'); sys.path.insert(0, '.')
import bench_pcie
bench_pcie.run(8192, 8192, 1, 16, True, reps=3)
\"" > $R/$out/mc.log 2>&1
cd $R; find $out/mc -name "*memory_copy_trace.csv" | head -1 | xargs -I{} cp {} $out/memory_copy_trace.csv; rm -rf $out/mc
