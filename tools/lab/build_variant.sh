#!/bin/bash
# Lab: build a variant of the library with extra flags on ONE source (default tr_gemm.hip) -> tools/lab/libtr_<name>.so
#   tools/lab/build_variant.sh <name> "<flags>" [source.hip]
set -e
name=$1; flags=$2; src=${3:-tr_gemm.hip}
cs=tokenreduction_amd/csrc
obj=/tmp/variant_${name}_${src%.hip}.o
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -Wno-inline-asm -ffp-contract=fast -fno-slp-vectorize $flags -c $cs/$src -o $obj
others=$(ls $cs/*.o | grep -v "/${src%.hip}.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $obj $others -o tools/lab/libtr_${name}.so
echo built tools/lab/libtr_${name}.so
