#!/bin/bash
# Round 5: N(x) = c1 + c2 x and D(x) = 1 + c3 x of the close PQ form as one FMA each (AG_PQ_ND_FMA=1; with AG_PQ_TAB_FORM=3 the tables hold
# c2 2^N and c3 2^N and the multiply by 2^N goes too): exact-match rates against the oracle / float64 truth, then the fresh-data A/B.
#   tools/ab_variants.sh write_kernels_p1,write_kernels_p2,write_kernels_p3,write_kernels_p32,write_kernels_p33,write_kernels_p36 "-DAG_PQ_ND_FMA=1" fma2 "-DAG_PQ_ND_FMA=1 -DAG_PQ_TAB_FORM=3" fma3
#   /usr/local/graft/bin/gpurun --timeout 900 -- 'bash tools/gpu/r05_pq_nd_fma.sh'
set -u
cd "$(dirname "$0")/../.."
O=gpurun_out/r05g; mkdir -p $O
F=$O/pq_nd_fma.txt; : > $F
for v in tree fma2 fma3; do
  lib=$PWD/avif-format_amd/variants/libavifgpu_$v.so; [ "$v" = tree ] && lib=$PWD/avif-format_amd/libavifgpu.so
  echo "== $v: exact-match rates" >> $F
  AVIFGPU_LIB=$lib timeout 600 python -m pytest tests/test_gpu_t2_truth.py tests/test_gpu_write.py::test_write_pq_code_boundaries -m gpu -q -s 2>&1 | grep -i "exact\|sweep\|passed\|failed" >> $F
done
ONLY="C4 8192^2 RGB f32 -> 10-bit PQ|D12 8192^2 RGB f32 -> 12-bit PQ|D12 8192^2 RGBA f32|C5 16384|GEO 7952x5304 RGB f32|D12 + ICC" bash tools/gpu/ab_fresh.sh tree fma2 fma3 >> $F 2>&1
cat $F
