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
# libbtcomm.so: the RCCL exchange steps (include/btcomm.h), a library of its own so that libbtgpu.so carries no RCCL dependency
comm_src="$here/comm/bt_comm.hip"
comm_out="$here/../libbtcomm.so"
if [ ! -f "$comm_out" ] || [ "$comm_src" -nt "$comm_out" ] || [ "$out" -nt "$comm_out" ] || [ -n "$(find "$here/../../include" -name '*.h' -newer "$comm_out" | head -1)" ]; then
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -Wall -Wno-unused-function "$comm_src" -o "$comm_out" -L"$here/.." -l:libbtgpu.so -lrccl -Wl,-rpath,'$ORIGIN'
fi
echo "built $comm_out"
