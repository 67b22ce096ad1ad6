#!/bin/bash
# Round 3, pass b: PQ variants (accuracy table + library A/B on the PQ rows), tile-height sweep of the hot kernel, u8 footprint A/B.
out=gpurun_out/r03b; mkdir -p $out
fmt='import sys,json
for l in sys.stdin:
    try: d=json.loads(l)
    except Exception: continue
    print("%-84s %.4f ms  %.3f  %s" % (d["config"][:84], d["ms_mean"], d["frac_of_8TBs"], d["kernel"][:50]))'
timeout 300 tools/pq_variants > $out/pq_variants.txt 2>&1
V=avif-format_amd/variants
for rep in 1 2; do
for lib in default pqhi0 pqhi2; do
  echo "== lib $lib (rep $rep)"
  if [ $lib = default ]; then unset AVIFGPU_LIB; else export AVIFGPU_LIB=$PWD/$V/libavifgpu_$lib.so; fi
  python tools/bench_configs.py "C4 8192^2 RGB f32 -> 10-bit PQ 4:4:4" "C5 16384" "C4 8192^2 RGB f32 -> 10-bit PQ 4:2:0" "W32 8192" "GEO 7952x5304 RGBA" "Gray32" "C5-like" 2>/dev/null | python -c "$fmt"
done; done > $out/pq_lib_ab.txt 2>&1
unset AVIFGPU_LIB
for lib in default pqhi0 pqhi2; do
  echo "== lib $lib"
  if [ $lib = default ]; then unset AVIFGPU_LIB; else export AVIFGPU_LIB=$PWD/$V/libavifgpu_$lib.so; fi
  python -m pytest tests/test_gpu_t2_truth.py -q -s -k "pq_write" 2>&1 | grep -E "PQ OETF|mismatches|passed|failed|Error"
done > $out/pq_t2.txt 2>&1
unset AVIFGPU_LIB
# hot kernel: 8 vs 4 pixels per lane on row tiles of the 8-way split (variant word: 7 = px8, 5 = px4)
for h in 512 1024 2048 4096 8192; do
  echo "== height $h"
  python bench.py --height $h --steps 400 --no-cpu-baseline --no-pcie --no-rotate --no-c5 --sweep 7,5,7,5 2>&1 >/dev/null | grep sweep
done > $out/tile_sweep.txt 2>&1
for rep in 1 2; do
for lib in default w8nc4; do
  echo "== lib $lib (rep $rep)"
  if [ $lib = default ]; then unset AVIFGPU_LIB; else export AVIFGPU_LIB=$PWD/$V/libavifgpu_$lib.so; fi
  python tools/bench_configs.py "C2" "W8 8192^2 RGB8" "GEO 7952x5304 RGB8" "BIG 16384^2 RGB8" 2>/dev/null | python -c "$fmt"
done; done > $out/w8nc_ab.txt 2>&1
unset AVIFGPU_LIB
python -m pytest tests -m gpu -x -q 2>&1 | tail -3 > $out/pytest.txt
cat $out/pq_variants.txt $out/pq_lib_ab.txt $out/pq_t2.txt $out/tile_sweep.txt $out/w8nc_ab.txt $out/pytest.txt
