#!/bin/bash
# Development: build library variants (tools/_abl/libesmi_<tag>.so) with extra -D flags; translation units compile in parallel
# and only the ones whose flags / sources changed are rebuilt.  usage: tools/build_variants.sh name1 "-DFLAG ..." name2 "..." ...
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
mkdir -p "$ROOT/tools/_abl"
while [ $# -gt 0 ]; do
  tag=$1; flags=$2; shift 2
  python - "$tag" "$flags" <<PY || echo "BUILD FAILED: $tag"
import sys; sys.path.insert(0, "$ROOT")
import __graft_entry__ as g
g.build_library("libesmi_%s.so" % sys.argv[1], sys.argv[2].split(), outdir="$ROOT/tools/_abl")
PY
done
ls -la "$ROOT"/tools/_abl/*.so
