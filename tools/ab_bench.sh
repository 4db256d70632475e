#!/bin/bash
# usage: tools/ab_bench.sh "<lib> <bench flags>" ...   (development: A/B variants on one box)
cp efficientspeech_amd/libesmi.so /tmp/libesmi_default.so
for i in 1 2; do for spec in "$@"; do
  set -- $spec; lib=$1; shift
  cp $lib efficientspeech_amd/libesmi.so
  python bench.py --no-cpu-baseline "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$spec', round(d['value']/1e8,3), round(d['ms_per_step'],4), round(d['roofline']['kernel_ms'],4))"
done; done
cp /tmp/libesmi_default.so efficientspeech_amd/libesmi.so
