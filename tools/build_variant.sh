#!/bin/bash
# build_variant.sh <name> "<extra hipcc flags>" [files...]: a copy of the product library with the given sources (default: those that
# include site_internal.h) recompiled with extra flags -> tools/lib/lib_<name>.so (select with ALIGNQ_SO; A/B tools only).
set -e
cd "$(dirname "$0")/../alignq_amd/csrc"
name=$1; extra=$2; shift 2
files=${@:-"bn_kernels.hip bnq_kernels.hip conv_kernels.hip corr_large_kernels.hip site1_kernels.hip site4_kernels.hip site_kernels.hip"}
out=../../tools/lib/var_$name; mkdir -p $out
FLAGS="-O3 -fPIC -std=c++17 --offload-arch=gfx950 -ffp-contract=off -Wall -Wno-unused-function"
objs=""
for f in *.hip; do
  o=../lib/${f%.hip}.o
  for v in $files; do if [ "$v" == "$f" ]; then o=$out/${f%.hip}.o; /opt/rocm/bin/hipcc $FLAGS $extra -c $f -o $o & fi; done
  objs="$objs $o"
done
wait
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 $objs -o ../../tools/lib/lib_$name.so
echo built tools/lib/lib_$name.so
