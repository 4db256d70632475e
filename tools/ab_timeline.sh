#!/bin/bash
# usage: tools/ab_timeline.sh <lib> ...   (development: per-kernel timeline of variants on one box)
cp efficientspeech_amd/libesmi.so /tmp/libesmi_default.so
for lib in "$@"; do
  cp $lib efficientspeech_amd/libesmi.so
  d=gpurun_out/tl_$(basename $lib .so); mkdir -p $d
  (cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$d -o r -- python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline > /dev/null 2>&1)
  echo $lib; python tools/step_timeline.py $d | grep -v "^sum\|fillBuffer\|length_reg" | awk '{printf "%s %s | ", $1, $(NF-1)} END {print ""}'
  python bench.py --no-cpu-baseline | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('   bench', round(d['value']/1e8,3), round(d['ms_per_step'],4), round(d['roofline']['kernel_ms'],4))"
done
cp /tmp/libesmi_default.so efficientspeech_amd/libesmi.so
