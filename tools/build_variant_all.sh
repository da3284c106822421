#!/bin/bash
# libbtgpu_<tag>.so: the library with extra flags for every translation unit that instantiates the Gibbs sweep (tuning experiments; load with
# BTGPU_LIB=bayestyper_amd/libbtgpu_<tag>.so).  usage: tools/build_variant_all.sh <tag> <flags...>
set -euo pipefail
root="$(cd "$(dirname "$0")/.." && pwd)"
tag=$1; shift
src="$root/bayestyper_amd/csrc"
mkdir -p "$root/build/variant_obj"
objs=()
pids=()
for s in "$src"/*.hip; do
  b="$(basename "${s%.hip}")"
  case "$b" in
    bt_gibbs|bt_gibbs_hot_kernel|bt_gibbs_single_kernel|bt_gibbs_simple_kernel|bt_gibbs_chain_kernel)
      o="$root/build/variant_obj/${b}_$tag.o"
      hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wno-unused-function "$@" -c "$s" -o "$o" &
      pids+=($!)
      objs+=("$o");;
    *) objs+=("$src/$b.o");;
  esac
done
for p in "${pids[@]}"; do wait "$p"; done
hipcc --offload-arch=gfx950 -shared -fPIC "${objs[@]}" -o "$root/bayestyper_amd/libbtgpu_$tag.so"
echo "built libbtgpu_$tag.so"
