#!/bin/bash
# Build A/B variants of libavifgpu.so that differ in one translation unit compiled with extra -D flags.
#   tools/ab_variants.sh read_kernels "-DAG_X=1" name1 "-DAG_X=2" name2 ...
#   tools/ab_variants.sh write_kernels,pattern_probe "-DAG_X=1" name1 ...      (several units rebuilt with the same flags)
# Results: avif-format_amd/variants/libavifgpu_<name>.so  (travels with gpurun; git-ignored as *.so)
set -e
cd "$(dirname "$0")/../avif-format_amd"
units=$1; shift
IFS=, read -ra U <<< "$units"
mkdir -p variants build
make -s libavifgpu.so
names=()
while [ $# -gt 1 ]; do
  flags=$1; name=$2; shift 2; names+=("$name")
  for unit in "${U[@]}"; do
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -Wall -Wno-unused-function $flags -c csrc/$unit.hip -o variants/$unit.$name.o &
  done
done
wait
for name in "${names[@]}"; do
  objs=""
  for f in avifgpu_api.hip pipeline.hip write_kernels.hip read_kernels.hip pattern_probe.hip host_shim.cpp host_decisions.cpp icc_profile.cpp; do
    o=build/$f.o
    for unit in "${U[@]}"; do [ "$f" = "$unit.hip" ] && o=variants/$unit.$name.o; done
    objs="$objs $o"
  done
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o variants/libavifgpu_$name.so $objs
  echo built variants/libavifgpu_$name.so
done
rm -f variants/*.o
