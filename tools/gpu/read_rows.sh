#!/bin/bash
# read rows of bench_configs with their twins, for one or more library builds:  tools/gpu/read_rows.sh tree name ...  (ROWS="pat|pat")
IFS="|" read -ra pats <<< "${ROWS:-R8 8192^2 8-bit 4:2|GEO 7952x5304 8-bit|D12 8192^2 12-bit 4:2:2|R32 8192^2 12-bit 4:2:0|BIG 16384^2 10-bit 4:2:0}"
for rep in 1 2; do for v in "$@"; do
  lib=$PWD/avif-format_amd/variants/libavifgpu_$v.so; [ "$v" = tree ] && lib=$PWD/avif-format_amd/libavifgpu.so
  echo "== $v (pass $rep)"
  AVIFGPU_LIB=$lib python tools/bench_configs.py "${pats[@]}" 2>/dev/null | python -c '
import sys,json
for l in sys.stdin:
    try: d=json.loads(l)
    except Exception: continue
    print("%-74s %.4f ms %.3f | twin %s ms %s  kernel/twin %s" % (d["config"][:74], d["ms_mean"], d["frac_of_8TBs"], d.get("twin_ms_mean"), d.get("twin_frac_of_8TBs"), d.get("frac_of_twin")))'
done; done
