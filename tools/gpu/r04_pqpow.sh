#!/bin/bash
# Round 4, second half: the transcendental-free first power of the PQ OETF (AG_PQ_TAB_FORM 4) -- what the packed clamp does on the
# device, exactness through the library, and speed against form 2 (variants built by tools/ab_variants.sh: form2, f4maxclamp, f4seg4).
mkdir -p gpurun_out/r04
{
echo "== v_pk_mul_f32 clamp"; tools/pkclamp_check
echo "== tests"
python -m pytest tests/test_gpu_extremes.py tests/test_gpu_t2_truth.py tests/test_gpu_fullsize.py -x -q -s 2>&1 | grep -E "PQ OETF|PQ sweep|exact 0|passed|failed|Error|error" | head -80
python -m pytest tests/test_gpu_write.py tests/test_gpu_kernel_equivalence.py -x -q 2>&1 | tail -3
python -m pytest tests/test_gpu_icc.py -x -q -s -k "mixed or sampled" 2>&1 | grep -E "icc-mixed|passed|failed|Error|error" | head -60
} > gpurun_out/r04/pqpow_tests.txt 2>&1
ONLY="C4 8192^2 RGB f32 -> 10-bit PQ 4:4:4|C4 8192^2 RGB f32 -> 10-bit PQ 4:2:0|C4 8192^2 RGB f32 -> 10-bit PQ interleaved|D12 8192^2 RGB f32 -> 12-bit PQ 4:2:2 nearest|D12 8192^2 RGB f32 -> 12-bit PQ 4:4:4|D12 8192^2 RGBA f32 -> 12-bit PQ 4:2:2|C5 16384|GEO 7952x5304 RGB f32 -> 10-bit PQ 4:2:0|sampled-curve doc profile -> Rec.2020) 8192|C4 + ICC (linear" \
  tools/gpu/ab_libs.sh form2 tree f4maxclamp f4seg4 > gpurun_out/r04/pqpow_ab.txt 2>&1
