#!/bin/bash
# Round 4 evidence files that are not part of the big table: PQ rows round-3 library vs this tree (interleaved, one box), the T2 sweeps
# through the library, the candidate table of tools/pq_variants, the issue counters of the C4 kernels.
out=gpurun_out/r04e; mkdir -p $out
ONLY="C4 8192^2 RGB f32 -> 10-bit PQ|D12 8192^2 RGB|C5 16384|W32 8192|C4 + ICC (linear|D12 + ICC|GEO 7952x5304 RGB f32|GEO 6001" tools/gpu/ab_libs.sh r03 tree > $out/pq_close_form_ab.txt 2>&1
timeout 300 python -m pytest tests/test_gpu_t2_truth.py tests/test_gpu_write.py::test_write_pq_code_boundaries tests/test_gpu_fullsize.py -m gpu -q -s 2>&1 | grep -i "exact\|sweep\|passed\|failed" > $out/t2_sweep_library.txt
timeout 200 tools/pq_variants > $out/pq_variants.txt 2>&1
timeout 300 tools/gpu/r04_pmc_pq.sh r03 tree > $out/pmc_pq_summary.txt 2>&1; cp gpurun_out/r04/pmc_pq_r03.json gpurun_out/r04/pmc_pq_tree.json $out/ 2>/dev/null
tail -30 $out/pq_close_form_ab.txt; cat $out/t2_sweep_library.txt | head -30
