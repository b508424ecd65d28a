#!/bin/bash
# One translation unit rebuilt with extra flags and linked with the shipped objects into lib/libdvae_hip_<name>.so (selected at
# run time with DVAE_HIP_LIB): the A/B builds of a GPU visit.      tools/build_variant.sh <name> <source stem> "<extra flags>"
set -e
cd "$(dirname "$0")/../disentangling-vae_amd"
name=$1; stem=$2; extra=$3
python build.py > /dev/null
obj=build/variant_${name}_${stem}.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function $extra -c csrc/$stem.hip -o $obj
objs=$(ls build/*.o | grep -v "/variant_" | grep -v "/$stem.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o lib/libdvae_hip_$name.so $objs $obj -ldl
echo lib/libdvae_hip_$name.so
