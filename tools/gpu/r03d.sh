#!/bin/bash
# Round 3, pass d: suite on the new tree (table-free reads, icc = 2 on the streaming kernels, split-exponent PQ); A/Bs of each.
out=gpurun_out/r03d; mkdir -p $out
fmt='import sys,json
for l in sys.stdin:
    try: d=json.loads(l)
    except Exception: continue
    print("%-84s %.4f ms  %.3f  %s" % (d["config"][:84], d["ms_mean"], d["frac_of_8TBs"], d["kernel"][:70]))'
python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > $out/pytest.txt
python -m pytest tests/test_gpu_icc.py -q -s -k "icc2_streaming" 2>&1 | grep -E "icc2-streaming|passed|failed" > $out/icc2_accuracy.txt
V=avif-format_amd/variants
for rep in 1 2; do
for lib in default arith0 arith12; do
  echo "== lib $lib (rep $rep)"
  if [ $lib = default ]; then unset AVIFGPU_LIB; else export AVIFGPU_LIB=$PWD/$V/libavifgpu_$lib.so; fi
  python tools/bench_configs.py "R8 " "R16 " "R32 " "BIG 16384^2 8-bit" "BIG 16384^2 10-bit" "GEO 7952x5304 8-bit" "GEO 6000x4000 10-bit" 2>/dev/null | python -c "$fmt"
done; done > $out/read_arith_ab.txt 2>&1
for rep in 1 2; do
for lib in default pqhix2 pqhi0; do
  echo "== lib $lib (rep $rep)"
  if [ $lib = default ]; then unset AVIFGPU_LIB; else export AVIFGPU_LIB=$PWD/$V/libavifgpu_$lib.so; fi
  python tools/bench_configs.py "C5 16384" "GEO 7952x5304 RGBA" "C5-like" 2>/dev/null | python -c "$fmt"
done; done > $out/pq_hi_ab.txt 2>&1
for rep in 1 2; do
for lib in default icc2hot0; do
  echo "== lib $lib (rep $rep)"
  if [ $lib = default ]; then unset AVIFGPU_LIB; else export AVIFGPU_LIB=$PWD/$V/libavifgpu_$lib.so; fi
  python tools/bench_configs.py "sRGB parametric" 2>/dev/null | python -c "$fmt"
done; done > $out/icc2_ab.txt 2>&1
unset AVIFGPU_LIB
python tools/gpu/pcie_pin_ab.py > $out/pcie_pin_ab.jsonl 2>&1
cat $out/pytest.txt $out/icc2_accuracy.txt $out/read_arith_ab.txt $out/pq_hi_ab.txt $out/icc2_ab.txt $out/pcie_pin_ab.jsonl
