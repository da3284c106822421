#!/bin/bash
# Build libbthost.so: the C++ host layer that mirrors the reference's class interface over the C ABI of libbtgpu.so.
set -euo pipefail
here="$(cd "$(dirname "$0")" && pwd)"
g++ -std=c++17 -O2 -fPIC -shared -Wall -I"$here/../../include" "$here"/*.cpp -o "$here/../libbthost.so" -L"$here/.." -l:libbtgpu.so -Wl,-rpath,'$ORIGIN' -lpthread -lz
echo "built $here/../libbthost.so"
