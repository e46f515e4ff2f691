#!/bin/bash
# tools/build_variant.sh NAME FILE.hip "-DFLAG=..." : links variants/libvecgo_NAME.so = the current objects with
# FILE.hip recompiled under the extra flags (kernel experiments: run a tool with VECGO_HIP_LIB=variants/...).
set -e
cd "$(dirname "$0")/../vecgo_amd/csrc"
name=$1; file=$2; flags=$3
mkdir -p ../../variants build_$name
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -fvisibility=hidden \
  -Wall -Wno-unused-function -I../../include $flags $( [ "$file" = k_pq_train.hip ] && echo "-mllvm -amdgpu-mfma-vgpr-form" ) -c $file -o build_$name/${file%.hip}.o
objs=$(ls build/*.o | grep -v "/${file%.hip}.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../variants/libvecgo_$name.so $objs build_$name/${file%.hip}.o
rm -rf build_$name
echo built variants/libvecgo_$name.so
