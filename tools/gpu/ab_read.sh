# A/B of read-kernel variants (built by tools/ab_variants.sh): same bench, one library per run
mkdir -p gpurun_out
: > gpurun_out/ab_read.jsonl
for v in ""; do
  if [ -z "$v" ]; then lib=""; name=base; else lib=$PWD/avif-format_amd/variants/libavifgpu_$v.so; name=$v; fi
  [ -n "$lib" ] && [ ! -f "$lib" ] && continue
  AVIFGPU_LIB=$lib python tools/bench_configs.py R8 R16 R32 2>/dev/null | sed "s/^{/{\"variant\": \"$name\", /" >> gpurun_out/ab_read.jsonl
done
python - <<'PY'
import json
rows=[json.loads(l) for l in open('gpurun_out/ab_read.jsonl')]
cfgs=[]; 
for r in rows:
    if r['config'] not in cfgs: cfgs.append(r['config'])
vs=[]
for r in rows:
    if r['variant'] not in vs: vs.append(r['variant'])
print("%-60s" % "config" + "".join("%9s" % v for v in vs))
for c in cfgs:
    print("%-60s" % c[:60] + "".join("%9.4f" % next((r['ms_mean'] for r in rows if r['config']==c and r['variant']==v), float('nan')) for v in vs))
PY
timeout 600 python -m pytest tests/test_gpu_read.py tests/test_gpu_tiles.py -m gpu -q -x 2>&1 | tail -3
