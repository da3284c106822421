#!/bin/bash
# libbtgpu_<tag>.so: the library with extra flags for ONE translation unit (tuning experiments; load with BTGPU_LIB=bayestyper_amd/libbtgpu_<tag>.so)
# usage: tools/build_variant.sh <tag> <unit without .hip> <flags...>
set -euo pipefail
root="$(cd "$(dirname "$0")/.." && pwd)"
tag=$1; unit=$2; shift 2
src="$root/bayestyper_amd/csrc"
mkdir -p "$root/build/variant_obj"
o="$root/build/variant_obj/${unit}_$tag.o"
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wno-unused-function "$@" -c "$src/$unit.hip" -o "$o"
objs=()
for s in "$src"/*.hip; do b="$(basename "${s%.hip}")"; if [ "$b" = "$unit" ]; then objs+=("$o"); else objs+=("$src/$b.o"); fi; done
hipcc --offload-arch=gfx950 -shared -fPIC "${objs[@]}" -o "$root/bayestyper_amd/libbtgpu_$tag.so"
echo "built libbtgpu_$tag.so"
