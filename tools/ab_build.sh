#!/bin/bash
# A/B build of ONE translation unit: tools/ab_build.sh <tag> <source.hip> <extra flags...>  ->  hands-on-point-cloud-processing_amd/libpcr_<tag>.so
# (the other objects are those of the last regular build; select the result with PCR_LIB_PATH, same ABI)
set -e
tag=$1; src=$2; shift 2
P=$(dirname "$0")/../hands-on-point-cloud-processing_amd
hipcc -O3 --offload-arch=gfx950 -std=c++17 -fPIC -ffp-contract=off -fno-slp-vectorize -mllvm -amdgpu-mfma-vgpr-form -fno-fast-math \
      -I"$P/../include" -I"$P/csrc" "$@" -c "$P/csrc/$src" -o "/tmp/ab_${tag}_$src.o"
objs=$(ls "$P"/build/*.o | grep -v "/$src.o")
hipcc --offload-arch=gfx950 -shared -fPIC -o "$P/libpcr_$tag.so" $objs "/tmp/ab_${tag}_$src.o" -ldl
echo "$P/libpcr_$tag.so"
