#!/bin/bash
# A/B builds of the kernels: tools/build_variant.sh NAME [extra hipcc flags...] compiles
# sift_kernels.hip with the extra flags and links it with the library's other objects into
# sift_amd/build/variants/NAME.so (select it with SIFT_AMD_LIB=<path> in the tools).
set -e
cd "$(dirname "$0")/.."
name=$1; shift
mkdir -p sift_amd/build/variants
FLAGS="--offload-arch=gfx950 -O3 -ffp-contract=off -fno-slp-vectorize -fPIC -std=c++17 -Wno-unused-result -Wno-unused-value"
hipcc $FLAGS "$@" -c sift_amd/csrc/sift_kernels.hip -o sift_amd/build/variants/$name.o
hipcc --offload-arch=gfx950 -shared -fPIC -o sift_amd/build/variants/$name.so sift_amd/build/variants/$name.o \
    sift_amd/build/sift_ctx.o sift_amd/build/sift_compat.o sift_amd/build/sift_match.o -Wl,-Bsymbolic-functions
echo sift_amd/build/variants/$name.so
