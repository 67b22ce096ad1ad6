#!/bin/bash
out=gpurun_out/r04; mkdir -p $out
ONLY="GEO 7952x5304 RGB f32|GEO 6000x4000 RGB f32|C4 8192^2 RGB f32 -> 10-bit PQ 4:4:4|C4 8192^2 RGB f32 -> 10-bit PQ 4:2:0" tools/gpu/ab_libs.sh tree edge_cached > $out/edge_cached_ab.txt 2>&1
cat $out/edge_cached_ab.txt
export PMC_GROUPS="FETCH_SIZE;WRITE_SIZE"
for v in tree edge_cached; do
  lib=$PWD/avif-format_amd/variants/libavifgpu_$v.so; [ "$v" = tree ] && lib=$PWD/avif-format_amd/libavifgpu.so
  AVIFGPU_LIB=$lib python tools/gpu/pmc_rows.py $out/pmc_edge_$v.json "GEO 7952x5304 RGB f32 -> 10-bit PQ 4:2:0" "GEO 7952x5304 RGB f32 -> 10-bit PQ 4:4:4" > /dev/null 2>&1
  python - <<PY
import json
d=json.load(open("$out/pmc_edge_$v.json"))
for k,v in d["rows"].items():
    rd=v.get("FETCH_SIZE",0)*2048; wr=v.get("WRITE_SIZE",0)*1024
    print("$v", k[:60], "traffic/algorithmic %.4f (read %.0f MB write %.0f MB)" % ((rd+wr)/v["algorithmic_bytes"], rd/1e6, wr/1e6))
PY
done
