#!/bin/bash
# Build A/B variants of libavifgpu.so that differ in one or more translation units compiled with extra flags.
#   tools/ab_variants.sh read_kernels "-DAG_X=1" name1 "-DAG_X=2" name2 ...
#   tools/ab_variants.sh write_kernels,pattern_probe "-DAG_X=1" name1 ...      (several units rebuilt with the same flags)
#   tools/ab_variants.sh write_kernels_p1 "-DAG_Y=1" name ...                  (ONE code object of write_kernels.hip: 1 streaming, 8 / 16 / 32 / 33 generic, 36 sampled ICC; read_kernels_p8 / 16 / 32 likewise -- seconds instead of minutes)
# "write_kernels" / "read_kernels" alone rebuild that file as one code object (AG_*_PART 0) standing in for all its parts.
# Results: avif-format_amd/variants/libavifgpu_<name>.so  (travels with gpurun; git-ignored as *.so)
set -e
cd "$(dirname "$0")/../avif-format_amd"
units=$1; shift
IFS=, read -ra U <<< "$units"
mkdir -p variants build
make -s -j8 libavifgpu.so
names=()
while [ $# -gt 1 ]; do
  flags=$1; name=$2; shift 2; names+=("$name")
  for unit in "${U[@]}"; do
    src=$unit; part=""
    if [[ "$unit" == write_kernels_p* ]]; then src=write_kernels; part="-DAG_WRITE_PART=${unit#write_kernels_p}"; fi
    if [[ "$unit" == read_kernels_p* ]]; then src=read_kernels; part="-DAG_READ_PART=${unit#read_kernels_p}"; fi
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -Wall -Wno-unused-function --offload-compress $part $flags -c csrc/$src.hip -o variants/$unit.$name.o &
  done
done
wait
base=$(make -s print-objs)
for name in "${names[@]}"; do
  objs=""
  for o in $base; do
    f=$(basename $o .o)                      # e.g. read_kernels.hip, write_kernels_p8.hip, host_shim.cpp
    keep=$o
    for unit in "${U[@]}"; do
      [ "$f" = "$unit.hip" ] && keep=variants/$unit.$name.o
      # write_kernels built as ONE code object stands in for all the parts
      if [ "$unit" = write_kernels ] && [[ "$f" == write_kernels_p*.hip ]]; then keep=""; fi
      if [ "$unit" = read_kernels ] && [[ "$f" == read_kernels_p*.hip ]]; then keep=""; fi
    done
    objs="$objs $keep"
  done
  for unit in "${U[@]}"; do if [ "$unit" = write_kernels ] || [ "$unit" = read_kernels ]; then objs="$objs variants/$unit.$name.o"; fi; done
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o variants/libavifgpu_$name.so $objs
  echo built variants/libavifgpu_$name.so
done
for name in "${names[@]}"; do for unit in "${U[@]}"; do rm -f variants/$unit.$name.o; done; done
