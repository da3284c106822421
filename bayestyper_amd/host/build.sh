#!/bin/bash
# Build libbthost.so (the C++ host layer that mirrors the reference's class interface over the C ABI of libbtgpu.so) and the
# `bayesTyper` executable (cluster / genotype command lines) on top of it.
set -euo pipefail
here="$(cd "$(dirname "$0")" && pwd)"
srcs=()
for f in "$here"/*.cpp; do b="$(basename "$f")"; [ "$b" = main.cpp ] || [ "$b" = tools_main.cpp ] || srcs+=("$f"); done
g++ -std=c++17 -O2 -fPIC -shared -Wall -I"$here/../../include" "${srcs[@]}" -o "$here/../libbthost.so" -L"$here/.." -l:libbtgpu.so -Wl,-rpath,'$ORIGIN' -lpthread -lz -ldl
echo "built $here/../libbthost.so"
g++ -std=c++17 -O2 -Wall -I"$here/../../include" "$here/main.cpp" -o "$here/../bayesTyper" -L"$here/.." -l:libbthost.so -l:libbtgpu.so -Wl,-rpath,'$ORIGIN' -lpthread -lz -ldl
echo "built $here/../bayesTyper"
g++ -std=c++17 -O2 -Wall -I"$here/../../include" "$here/tools_main.cpp" -o "$here/../bayesTyperTools" -L"$here/.." -l:libbthost.so -l:libbtgpu.so -Wl,-rpath,'$ORIGIN' -lpthread -lz -ldl
echo "built $here/../bayesTyperTools"
