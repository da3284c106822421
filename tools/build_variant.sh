#!/bin/bash
# usage: tools/build_variant.sh <name> [extra hipcc flags]  ->  bayestyper_amd/libbtgpu_<name>.so (load it with BTGPU_LIB=...); tuning experiments only
set -euo pipefail
root="$(cd "$(dirname "$0")/.." && pwd)"
name=$1; shift
src="$root/bayestyper_amd/csrc"
obj="$root/scratch/var_${name}_obj"
mkdir -p "$obj"
objs=()
for s in "$src"/*.hip; do
  o="$obj/$(basename "${s%.hip}").o"
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wno-unused-function -I"$root/include" "$@" -c "$s" -o "$o" &
  objs+=("$o")
done
wait
hipcc --offload-arch=gfx950 -shared -fPIC "${objs[@]}" -o "$root/bayestyper_amd/libbtgpu_${name}.so"
echo "built $root/bayestyper_amd/libbtgpu_${name}.so"
