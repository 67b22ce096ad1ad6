#!/bin/bash
# Round 3, pass u: scheduler knob test + 16-bit table with 96-byte records (A/B against the 128-byte ones), parity first
out=gpurun_out/r03u; mkdir -p $out
timeout 900 python -m pytest tests/test_gpu_multidevice.py -m gpu -q -x 2>&1 | tail -3 | tee $out/pytest_sched.txt
AVIFGPU_LIB=$PWD/avif-format_amd/variants/libavifgpu_rec96.so timeout 900 python -m pytest tests/test_icc16.py tests/test_icc_golden.py -m gpu -q -x 2>&1 | tail -3 | tee $out/pytest_rec96.txt
bash tools/gpu/icc16_ab.sh rec96 2>&1 | grep -v passed > $out/rec96_ab.txt; cat $out/rec96_ab.txt
