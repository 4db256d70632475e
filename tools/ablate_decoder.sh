#!/bin/bash
# Development tool: build ablated variants of libesmi.so (mel decoder only matters) for tools/bench_decoder.py.
set -e
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
OUT="$ROOT/tools/_abl"; mkdir -p "$OUT"
build() { tag=$1; shift; hipcc --offload-arch=gfx950 -O2 -std=c++17 -shared -fPIC "$@" -o "$OUT/libesmi_$tag.so" "$ROOT/efficientspeech_amd/csrc/esmi_abi.hip" & }
build base
build fasttanh -DESMI_DEC_TANH=tanh_fast_f32
build notanh -DESMI_DEC_TANH=ident_f32
build noln -DESMI_ABL_NO_LN
build notanh_noln -DESMI_DEC_TANH=ident_f32 -DESMI_ABL_NO_LN
build nobload -DESMI_ABL_NO_BLOAD
build nodw -DESMI_ABL_NO_DW
build mfmaonly -DESMI_DEC_TANH=ident_f32 -DESMI_ABL_NO_LN -DESMI_ABL_NO_BLOAD -DESMI_ABL_NO_DW
wait
ls -la "$OUT"
