#!/bin/bash
# usage: tools/pmc_pass.sh <tag> "<counter group>" ["<counter group>" ...]   (run on the GPU box via gpurun)
TAG=$1; shift
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for grp in "$@"; do
  name=pmc_$(echo $grp | cut -d' ' -f1)
  timeout 300 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $OUT/$name -o r -- python $REPO/bench.py --steps 3 --warmup 2 --no-cpu-baseline > $OUT/$name.log 2>&1
done
cd $REPO; python tools/prof_summary.py $OUT > /dev/null 2>&1
python - <<PY
import json
d=json.load(open("$OUT/pmc_counters.json"))["per_kernel_mean_per_launch"]
for k,v in d.items():
    print(k, {c:round(x) for c,x in v.items()})
PY
