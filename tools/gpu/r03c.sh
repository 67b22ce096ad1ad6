#!/bin/bash
# Round 3, pass c: the GPU suite on the new tree (PQ hi form, topology, packed u8 read math), the t2 sweep, the u8 read A/Bs (packed
# pair math, grid caps, workgroup sizes), the bench line with peak_measured.
out=gpurun_out/r03c; mkdir -p $out
fmt='import sys,json
for l in sys.stdin:
    try: d=json.loads(l)
    except Exception: continue
    print("%-84s %.4f ms  %.3f  %s" % (d["config"][:84], d["ms_mean"], d["frac_of_8TBs"], d["kernel"][:50]))'
python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > $out/pytest.txt
python -m pytest tests/test_gpu_t2_truth.py tests/test_topology.py -q -s -m gpu 2>&1 | grep -E "PQ OETF|mismatches|topology|passed|failed|Error" > $out/t2_topology.txt
V=avif-format_amd/variants
for rep in 1 2; do
for lib in default r8pk0 rcap1280 rcap2k rcap4k rblk128 rblk512; do
  echo "== lib $lib (rep $rep)"
  if [ $lib = default ]; then unset AVIFGPU_LIB; else export AVIFGPU_LIB=$PWD/$V/libavifgpu_$lib.so; fi
  python tools/bench_configs.py "R8 8192" "BIG 16384^2 8-bit" "GEO 7952x5304 8-bit" "R16 8192^2 12-bit mono" "R32 8192^2 10-bit 4:4:4" 2>/dev/null | python -c "$fmt"
done; done > $out/read_ab.txt 2>&1
unset AVIFGPU_LIB
python tools/bench_configs.py "C5 16384" "C4 8192^2 RGB f32 -> 10-bit PQ 4:4:4" "GEO 7952x5304 RGBA" "W32" 2>/dev/null | python -c "$fmt" > $out/pq_rows.txt
python bench.py > $out/bench.json 2> $out/bench.err
cat $out/pytest.txt $out/t2_topology.txt $out/read_ab.txt $out/pq_rows.txt $out/bench.json; tail -3 $out/bench.err
