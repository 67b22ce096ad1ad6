#!/bin/bash
# Round 3, pass k: the 16-bit table kernel in its v_dot2 form -- parity against the real lcms2, then speed beside the round-2 form
out=gpurun_out/r03k; mkdir -p $out
bash tools/gpu/icc16_ab.sh icc16r2 > $out/icc16_ab.txt 2>&1; cat $out/icc16_ab.txt
timeout 900 python -m pytest tests -m gpu -q -x -k "icc or golden or equivalence" 2>&1 | tail -3 | tee $out/pytest_icc.txt
