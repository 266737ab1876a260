#!/bin/sh
# Build the host-emulator flavour of the kernel library (test infrastructure).
set -e
HERE=$(cd "$(dirname "$0")" && pwd)
ROOT=$(cd "$HERE/../.." && pwd)
CXX=${CXX:-/opt/rocm/lib/llvm/bin/clang++}
[ -x "$CXX" ] || CXX=clang++
mkdir -p "$HERE/_build"
"$CXX" -std=c++17 -O2 -fPIC -shared -DSTCAT_EMU -x c++ -I"$HERE" -I"$ROOT/stcat_amd/csrc" \
  -Wno-unknown-attributes -Wno-unused-value \
  "$HERE/emu_main.cpp" -o "$HERE/_build/libstcat_emu.so" -lpthread
echo "$HERE/_build/libstcat_emu.so"
