#!/bin/bash
# Build libesmi_sim.so: the unmodified kernel sources compiled for the CPU wave simulator (tests only).
set -e
HERE="$(cd "$(dirname "$0")" && pwd)"
ROOT="$(cd "$HERE/../.." && pwd)"
CXX=${CXX:-/opt/rocm/lib/llvm/bin/clang++}
mkdir -p "$HERE/_build"
"$CXX" -x c++ -std=c++17 -O2 -g0 -DESMI_WAVESIM -I"$HERE" -I"$ROOT/efficientspeech_amd/csrc" \
    -fPIC -shared -Wno-unused-value -Wno-pass-failed \
    "$ROOT/efficientspeech_amd/csrc/esmi_abi.hip" "$HERE/wavesim.cpp" \
    -o "$HERE/_build/libesmi_sim.so" -lpthread
echo "built $HERE/_build/libesmi_sim.so"
