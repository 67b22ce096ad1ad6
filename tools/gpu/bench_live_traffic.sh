#!/bin/bash
# bench.py with the live PMC traffic measurement: the line, how long the whole command took, and the child mode alone
out=gpurun_out/live_traffic; mkdir -p $out
t0=$(date +%s.%N); python bench.py > $out/bench.json 2> $out/bench.err; t1=$(date +%s.%N)
python - <<PY
import json
d=json.load(open("$out/bench.json"))
r=d["roofline"]
print("wall s:", round($t1-$t0,1), "value", d["value"], "frac", r["frac"], "traffic", r.get("traffic"), r.get("traffic_over_algorithmic"))
print(r.get("traffic_source")); print(r.get("traffic_live_note"))
PY
tail -3 $out/bench.err
