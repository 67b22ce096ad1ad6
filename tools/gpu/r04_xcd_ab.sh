#!/bin/bash
# XCD-contiguous pieces (device_math.h, xcd_block) against piece = blockIdx.x (variants/libavifgpu_noxcd.so: the three kernel files
# built with -DAG_XCD_GROUPS=0): times of the rows whose spans share lines across workgroups and of the headline rows, interleaved
# twice, then HBM traffic of the GEO rows under both libraries (FETCH_SIZE / WRITE_SIZE in passes of their own).
mkdir -p gpurun_out/r04
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
ONLY="GEO|C4 8192^2 RGB f32 -> 10-bit PQ 4:4:4|C4 8192^2 RGB f32 -> 10-bit PQ 4:2:0|D12 8192^2 RGB f32 -> 12-bit PQ 4:2:2 nearest|C2' 8192|R8 8192^2 8-bit 4:2:0 BT.709|R8 8192^2 8-bit 4:2:2|D12 8192^2 12-bit 4:2:2|BIG 16384^2 RGB f32 -> 10-bit PQ 4:4:4|BIG 16384^2 10-bit 4:2:0|BIG 16384^2 8-bit|C5 16384|C3 8192|W16 8192^2 RGB16 -> 12-bit 4:2:0" \
  tools/gpu/ab_libs.sh noxcd run2 tree run8 > gpurun_out/r04/xcd_groups_ab.txt 2>&1
export BENCH_TWIN=0
for v in noxcd run2 tree run8; do
  lib=$R/avif-format_amd/variants/libavifgpu_$v.so
  [ "$v" = tree ] && lib=$R/avif-format_amd/libavifgpu.so
  out=$R/gpurun_out/r04/xcd_$v
  mkdir -p $out
  cd /tmp
  AVIFGPU_LIB=$lib timeout 300 rocprofv3 --kernel-trace -d $out/kt -o kt --output-format csv -- bash -c "cd $R && python tools/bench_configs.py GEO 2>/dev/null > $out/configs.jsonl" > $out/kt.log 2>&1
  for c in FETCH_SIZE WRITE_SIZE; do
    d=$(echo $c | tr A-Z a-z)
    AVIFGPU_LIB=$lib timeout 300 rocprofv3 --pmc $c --kernel-include-regex "write_|read_px" -d $out/$d -o p --output-format csv -- bash -c "cd $R && python tools/bench_configs.py GEO > /dev/null 2>&1" > $out/$d.log 2>&1
  done
  cd $R
  echo "== $v" >> gpurun_out/r04/xcd_groups_traffic.txt
  python tools/summarize_pmc.py $out $out/configs.jsonl $out/traffic.json >> gpurun_out/r04/xcd_groups_traffic.txt 2>&1
  rm -rf $out/kt $out/fetch_size $out/write_size
done
