#!/bin/bash
# Round 3, pass l: packed f32 stage B of the u8-plane kernels (AG_W8_PKF32: 0 never / 1 behind ICC / 2 always) + the mad chain of the
# 8-bit matrix-shaper: parity of the in-tree library, then speed of every row with u8 planes for the three builds, twice.
out=gpurun_out/r03l; mkdir -p $out
timeout 900 python -m pytest tests -m gpu -q -x -k "icc or golden or equivalence or write" 2>&1 | tail -3 | tee $out/pytest.txt
AVIFGPU_LIB=$PWD/avif-format_amd/variants/libavifgpu_pk2.so timeout 900 python -m pytest tests -m gpu -q -x -k "equivalence or write_parity or golden" 2>&1 | tail -3 | tee $out/pytest_pk2.txt
for rep in 1 2; do
for v in "" pk0 pk2; do
  echo "== ${v:-in-tree}"
  if [ -n "$v" ]; then export AVIFGPU_LIB=$PWD/avif-format_amd/variants/libavifgpu_$v.so; else unset AVIFGPU_LIB; fi
  python tools/bench_configs.py "8-bit" "RGBA8" "RGB8" 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    try: d=json.loads(l)
    except Exception: continue
    if d['kernel'].startswith('write'): print('%-84s %.4f ms  %.3f' % (d['config'][:84], d['ms_mean'], d['frac_of_8TBs']))"
done; done > $out/pk_ab.txt 2>&1
cat $out/pk_ab.txt
