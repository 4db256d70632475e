#!/bin/bash
# Build libesmi_sim.so: the unmodified kernel sources compiled for the CPU wave simulator (tests only).
# Every translation unit of efficientspeech_amd/csrc is compiled by the host clang++ (-DESMI_WAVESIM), in parallel, then linked.
set -e
HERE="$(cd "$(dirname "$0")" && pwd)"
ROOT="$(cd "$HERE/../.." && pwd)"
CXX=${CXX:-/opt/rocm/lib/llvm/bin/clang++}
OBJ="$HERE/_build/obj"
mkdir -p "$OBJ"
FLAGS="-std=c++17 -O2 -g0 -DESMI_WAVESIM -DESMI_RANGE_CHECK=1 -I$HERE -I$ROOT/efficientspeech_amd/csrc -fPIC -Wno-unused-value -Wno-pass-failed"
pids=()
for src in "$ROOT"/efficientspeech_amd/csrc/*.hip "$HERE/wavesim.cpp"; do
    "$CXX" -x c++ $FLAGS -c "$src" -o "$OBJ/$(basename "$src").o" &
    pids+=($!)
done
for p in "${pids[@]}"; do wait "$p"; done
"$CXX" -shared -fPIC "$OBJ"/*.o -o "$HERE/_build/libesmi_sim.so" -lpthread
echo "built $HERE/_build/libesmi_sim.so"
