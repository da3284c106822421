#!/bin/bash
# libbtgpu_prof.so: the same sources with the per-phase cycle counters compiled in (-DBT_PROF); load it with BTGPU_LIB=bayestyper_amd/libbtgpu_prof.so
set -euo pipefail
root="$(cd "$(dirname "$0")/.." && pwd)"
src="$root/bayestyper_amd/csrc"
obj="$root/build/prof_obj"
mkdir -p "$obj"
objs=()
for s in "$src"/*.hip; do
  o="$obj/$(basename "${s%.hip}").o"
  if [ ! -f "$o" ] || [ "$s" -nt "$o" ] || [ -n "$(find "$src" "$root/include" -name '*.h*' -newer "$o" | head -1)" ]; then
    hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wno-unused-function -DBT_PROF "$@" -c "$s" -o "$o"
  fi
  objs+=("$o")
done
hipcc --offload-arch=gfx950 -shared -fPIC "${objs[@]}" -o "$root/bayestyper_amd/libbtgpu_prof.so"
echo "built $root/bayestyper_amd/libbtgpu_prof.so"
