#!/bin/bash
# Round 3, pass n: whole GPU suite on the current tree, then the default bench line
out=gpurun_out/r03n; mkdir -p $out
timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | tail -4 | tee $out/pytest.txt
python bench.py > $out/bench.json 2> $out/bench.err; cat $out/bench.json; tail -2 $out/bench.err
