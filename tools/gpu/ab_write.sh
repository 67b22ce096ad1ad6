# A/B of write-kernel variants (built by tools/ab_variants.sh): same bench, one library per run
mkdir -p gpurun_out
: > gpurun_out/ab_write.jsonl
for v in ""; do
  if [ -z "$v" ]; then lib=""; name=base; else lib=$PWD/avif-format_amd/variants/libavifgpu_$v.so; name=$v; fi
  [ -n "$lib" ] && [ ! -f "$lib" ] && continue
  AVIFGPU_LIB=$lib python tools/bench_configs.py "C2" "W8" "W16" "RGBA8" "C3" "C4" "C5" "Gray" 2>/dev/null | sed "s/^{/{\"variant\": \"$name\", /" >> gpurun_out/ab_write.jsonl
done
python - <<'PY'
import json
rows=[json.loads(l) for l in open('gpurun_out/ab_write.jsonl')]
cfgs=[]; vs=[]
for r in rows:
    if r['config'] not in cfgs: cfgs.append(r['config'])
    if r['variant'] not in vs: vs.append(r['variant'])
print("%-60s" % "config" + "".join("%16s" % v for v in vs))
for c in cfgs:
    print("%-60s" % c[:60] + "".join("%9.4f|%.3f" % next(((r['ms_mean'],r['frac_of_8TBs']) for r in rows if r['config']==c and r['variant']==v), (float('nan'),0)) + " " for v in vs))
PY
timeout 600 python -m pytest tests/test_gpu_write.py tests/test_gpu_read.py tests/test_gpu_tiles.py -m gpu -q -x 2>&1 | tail -3
