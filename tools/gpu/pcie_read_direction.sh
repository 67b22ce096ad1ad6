#!/bin/bash
# Round 3, pass s: the open direction through host pointers (planes up, rows down) with and without the upload order; shim floor with the 16 MiB cap
out=gpurun_out/r03s; mkdir -p $out
for depth in 1 0; do for chunk in 16 32; do
AVIFGPU_UPLOAD_DEPTH=$depth $( [ $depth = 0 ] && echo AVIFGPU_SLOTS=4 ) timeout 200 python - 2>/dev/null <<PY
import sys
sys.path.insert(0, "tools"); sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import bench_pcie
print("upload_depth $depth")
bench_pcie.run_read(8192, 8192, $chunk, True, reps=6)
bench_pcie.run_read(8192, 8192, $chunk, False, reps=4)
PY
done; done | tee $out/read_direction.txt
timeout 300 python tools/bench_host_shim.py floor 2>/dev/null | cut -c1-330 | tee $out/shim_floor.txt
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; cd /tmp
timeout 300 rocprofv3 --memory-copy-trace -d $R/$out/mc -o mc --output-format csv -- bash -c "cd $R && python -c \"
import sys
sys.path.insert(0, 'tools'); sys.path.insert(0, 'tests'); sys.path.insert(0, '.')
import bench_pcie
bench_pcie.run_read(8192, 8192, 16, True, reps=3)
\"" > $R/$out/mc.log 2>&1
cd $R; find $out/mc -name "*memory_copy_trace.csv" | head -1 | xargs -I{} cp {} $out/memory_copy_trace_read.csv; rm -rf $out/mc
