#!/bin/bash
# Round 3, pass g: flat launches -- suite, then A/B on the GEO rows and a few regular ones; 2-rank rehearsal of bench.py on one GPU.
out=gpurun_out/r03g; mkdir -p $out
fmt='import sys,json
for l in sys.stdin:
    try: d=json.loads(l)
    except Exception: continue
    print("%-84s %.4f ms  %.3f  %s" % (d["config"][:84], d["ms_mean"], d["frac_of_8TBs"], d["kernel"][:90]))'
python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > $out/pytest.txt
V=avif-format_amd/variants
for rep in 1 2; do
for lib in default flat0 rflat0; do
  echo "== lib $lib (rep $rep)"
  if [ $lib = default ]; then unset AVIFGPU_LIB; else export AVIFGPU_LIB=$PWD/$V/libavifgpu_$lib.so; fi
  python tools/bench_configs.py "GEO" "C4 8192^2 RGB f32 -> 10-bit PQ 4:4:4" "C3 8192" "C5 16384" "R16 8192^2 12-bit mono" "R32 8192^2 10-bit 4:4:4" "R8 8192^2 8-bit 4:4:4" 2>/dev/null | python -c "$fmt"
done; done > $out/flat_ab.txt 2>&1
unset AVIFGPU_LIB
AVIFGPU_BENCH_SHARE_DEVICE=1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 50 --warmup 5 --no-cpu-baseline > $out/bench_2rank_rehearsal.json 2> $out/bench_2rank_rehearsal.err
cat $out/pytest.txt $out/flat_ab.txt; cat $out/bench_2rank_rehearsal.json | cut -c1-1500; tail -3 $out/bench_2rank_rehearsal.err
