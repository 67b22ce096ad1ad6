#!/bin/bash
# Round 5, item 1: every number on FRESH data (buffer sets rotating, > 1 GB between two visits of an address), and round 4's cache-policy
# decisions taken again under it.  Variants (tools/ab_variants.sh): e0 = every span load non-temporal, efirst = first load allocates,
# tree = AG_EDGE_CACHED as committed; rnt1 / rnt0 = every / no plane load of the opens non-temporal, tree = read_nt_loads() as committed.
# Results: gpurun_out/r05a/.
out=gpurun_out/r05a; mkdir -p $out
V=$PWD/avif-format_amd/variants
B="--steps 200 --warmup 20 --no-cpu-baseline --no-pcie --no-c5 --no-live-traffic --no-cold"
for rep in 1 2; do
  for v in tree e0 efirst; do
    lib=$V/libavifgpu_$v.so; [ $v = tree ] && lib=$PWD/avif-format_amd/libavifgpu.so
    AVIFGPU_LIB=$lib python bench.py $B > $out/bench_${v}_$rep.json 2> $out/bench_${v}_$rep.err
    python - $out/bench_${v}_$rep.json $v $rep <<'PY'
import json,sys
d=json.load(open(sys.argv[1])); r=d["roofline"]
print("bench %-7s pass %s  fresh %.4f ms frac %.4f | same buffers %.4f | twin fresh %.1f GB/s  frac_of_measured %.4f" % (sys.argv[2], sys.argv[3], r["kernel_ms_mean"], r["frac"], r.get("frac_same_buffers", 0), r.get("peak_measured", 0), r.get("frac_of_measured", 0)))
PY
  done
done 2>&1 | tee $out/headline_ab.txt
W='C4 8192^2 RGB f32 -> 10-bit PQ 4:4:4|C4 8192^2 RGB f32 -> 10-bit PQ 4:2:0|D12 8192^2 RGB f32 -> 12-bit PQ 4:2:2|GEO 7952x5304 RGB f32 -> 10-bit PQ 4:2:0|GEO 6001x4001|BIG 16384^2 RGB f32 -> 10-bit PQ 4:|D12 + ICC (linear Display-P3 doc -> Rec.2020) 8192^2 RGB f32 -> 12-bit PQ 4:2:2'
R='R8 8192^2 8-bit 4:2:0 BT.709|R8 8192^2 8-bit 4:2:2|R8 8192^2 8-bit 4:2:0 BT.601 + alpha|D12 8192^2 12-bit 4:2:2 BT.2020 PQ|HLG|BIG 16384^2 10-bit|BIG 16384^2 8-bit 4:2:0 BT.709 ->|GEO 7952x5304 8-bit|R16 8192^2 10-bit 4:4:4'
ab() {   # ab "patterns" lib...
  IFS="|" read -ra pats <<< "$1"; shift
  for rep in 1 2; do for v in "$@"; do
    lib=$V/libavifgpu_$v.so; [ $v = tree ] && lib=$PWD/avif-format_amd/libavifgpu.so
    echo "== $v (pass $rep)"
    AVIFGPU_LIB=$lib BENCH_TWIN=0 python tools/bench_configs.py "${pats[@]}" 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    try: d=json.loads(l)
    except Exception: continue
    print('%-86s fresh %.4f ms %.3f | same %.4f ms %.3f | sets %d' % (d['config'][:86], d['ms_mean'], d['frac_of_8TBs'], d.get('ms_same',0), d.get('frac_same',0), d['sets']))"
  done; done
}
ab "$W" tree e0 efirst > $out/write_policy_ab.txt 2>&1
ab "$R" tree rnt1 rnt0 > $out/read_policy_ab.txt 2>&1
python tools/bench_configs.py > $out/bench_configs_tree.jsonl 2> $out/bench_configs_tree.err
tail -3 $out/bench_configs_tree.err
cat $out/write_policy_ab.txt | head -80
cat $out/read_policy_ab.txt | head -80
wc -l $out/bench_configs_tree.jsonl
