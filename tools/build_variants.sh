#!/bin/bash
# Development: build library variants in parallel.  usage: tools/build_variants.sh name1 "-DFLAG ..." name2 "..." ...
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
OUT="$ROOT/tools/_abl"; mkdir -p "$OUT"
while [ $# -gt 0 ]; do
  tag=$1; flags=$2; shift 2
  ( /opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 -std=c++17 -Wno-pass-failed -shared -fPIC $flags -o "$OUT/libesmi_$tag.so" "$ROOT/efficientspeech_amd/csrc/esmi_abi.hip" 2> "$OUT/$tag.log" || echo "BUILD FAILED: $tag" ) &
done
wait
ls -la "$OUT"/*.so
