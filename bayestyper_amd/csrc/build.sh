#!/bin/bash
# Build libbtgpu.so (HIP, gfx950 only) in-tree.  Usage: build.sh [extra hipcc flags]
set -euo pipefail
here="$(cd "$(dirname "$0")" && pwd)"
out="$here/../libbtgpu.so"
srcs=("$here"/*.hip)
objs=()
# objects are rebuilt when a source / header is newer or when the extra flags differ from the previous build
flags="$*"
stamp="$here/.build_flags"
if [ ! -f "$stamp" ] || [ "$(cat "$stamp")" != "$flags" ]; then rm -f "$here"/*.o; echo "$flags" > "$stamp"; fi
for s in "${srcs[@]}"; do
  o="${s%.hip}.o"
  if [ ! -f "$o" ] || [ "$s" -nt "$o" ] || [ -n "$(find "$here" "$here/../../include" -name '*.h*' -newer "$o" 2>/dev/null | head -1)" ]; then
    hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wall -Wno-unused-function "$@" -c "$s" -o "$o"
  fi
  objs+=("$o")
done
hipcc --offload-arch=gfx950 -shared -fPIC "${objs[@]}" -o "$out"
echo "built $out"
