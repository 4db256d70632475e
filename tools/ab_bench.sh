#!/bin/bash
# usage: tools/ab_bench.sh <lib> [<lib> ...]   (development: A/B library variants on one box, two rounds, default bench flags)
cp efficientspeech_amd/libesmi.so /tmp/libesmi_default.so
for i in 1 2; do for lib in "$@"; do
  cp $lib efficientspeech_amd/libesmi.so
  python bench.py --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$lib', round(d['value']/1e8,3), round(d['ms_per_step'],4), round(d['roofline']['kernel_ms'],4))"
done; done
cp /tmp/libesmi_default.so efficientspeech_amd/libesmi.so
