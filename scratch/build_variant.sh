#!/bin/bash
# build a variant of the library with extra -D flags into tools/lib/lib_<name>.so:   scratch/build_variant.sh <name> <flags...>
set -e
name=$1; shift
out=tools/lib/var_$name
mkdir -p $out
cd alignq_amd/csrc
SRCS="common.hip quant_kernels.hip admm_sgd_kernels.hip site_kernels.hip site1_kernels.hip site4_kernels.hip corr_xy_kernels.hip corr_large_kernels.hip bn_kernels.hip bnq_kernels.hip multi_tensor_kernels.hip conv_kernels.hip head_kernels.hip"
for f in $SRCS; do
  /opt/rocm/bin/hipcc -O3 -fPIC -std=c++17 --offload-arch=gfx950 -ffp-contract=off -Wno-unused-function "$@" -c $f -o ../../$out/${f%.hip}.o &
done
wait
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 $(for f in $SRCS; do echo ../../$out/${f%.hip}.o; done) -o ../../tools/lib/lib_$name.so
echo built tools/lib/lib_$name.so
