#!/bin/bash
# Development (GPU box): time mel_decoder_kernel of several library variants, two rounds.  usage: tools/ab_decoder.sh [--h0] tag1 tag2 ...
ARGS=""; if [ "$1" == "--h0" ]; then ARGS="--h0"; shift; fi
LIBS=""; for t in "$@"; do LIBS="$LIBS tools/_abl/libesmi_$t.so"; done
for i in 1 2; do python tools/bench_decoder.py $ARGS --libs $LIBS 2>&1 | grep -v amdgpu.ids; done
