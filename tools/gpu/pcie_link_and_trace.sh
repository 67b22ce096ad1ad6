#!/bin/bash
# Round 3, pass o: the PCIe-inclusive path against the link's own ceiling -- plain copy patterns, then the library with its tile trace
out=gpurun_out/r03o; mkdir -p $out
python tools/pcie_patterns.py > $out/pcie_patterns.jsonl 2> $out/pcie_patterns.err; cat $out/pcie_patterns.jsonl; tail -2 $out/pcie_patterns.err
for lanes in 1 2; do for chunk in 8 32; do
AVIFGPU_LANES=$lanes timeout 120 python - 2>/dev/null <<PY
import sys
sys.path.insert(0, "tools"); sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import bench_pcie
bench_pcie.run(8192, 8192, 1, $chunk, True)
PY
done; done | tee $out/library.jsonl
AVIFGPU_TRACE=1 AVIFGPU_LANES=2 timeout 120 python - > $out/trace.txt 2>&1 <<'PY'
import sys
sys.path.insert(0, "tools"); sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import bench_pcie
bench_pcie.run(8192, 8192, 1, 8, True, reps=2)
PY
grep -c "avifgpu trace" $out/trace.txt; grep "avifgpu trace" $out/trace.txt | tail -100 | head -12
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; cd /tmp
timeout 300 rocprofv3 --memory-copy-trace --kernel-trace -d $R/$out/mc -o mc --output-format csv -- bash -c "cd $R && AVIFGPU_LANES=2 python -c \"
import sys
sys.path.insert(0, 'tools'); sys.path.insert(0, 'tests'); sys.path.insert(0, '.')
import bench_pcie
bench_pcie.run(8192, 8192, 1, 8, True, reps=3)
\"" > $R/$out/mc.log 2>&1
cd $R; find $out/mc -name "*memory_copy_trace.csv" | head -1 | xargs -I{} sh -c 'head -3 {}; wc -l {}; cp {} '$out'/memory_copy_trace.csv'; find $out/mc -name "*kernel_trace.csv" | head -1 | xargs -I{} cp {} $out/kernel_trace.csv; rm -rf $out/mc
