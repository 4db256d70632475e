#!/bin/bash
# Run on the GPU box (via gpurun): kernel-trace stats pass + one PMC pass per counter group.
# usage: tools/collect_profiles.sh <tag> [bench args...]   -> gpurun_out/<tag>/{stats,pmc_*}/ + summaries
set -u
TAG=${1:-r01}; shift || true
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
python $REPO/bench.py --no-cpu-baseline "$@" > $OUT/bench_unprofiled.json 2> $OUT/bench_unprofiled.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o r -- \
    python $REPO/bench.py --steps 10 --warmup 3 --event-every 1 --no-cpu-baseline "$@" > $OUT/stats.log 2>&1
for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE" \
           "SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_MFMA SQ_INSTS_VALU SQ_INSTS_LDS" "SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY"; do
  name=pmc_$(echo $grp | cut -d' ' -f1)
  timeout 300 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $OUT/$name -o r -- \
      python $REPO/bench.py --steps 3 --warmup 2 --no-cpu-baseline "$@" > $OUT/$name.log 2>&1
done
cd $REPO
python tools/prof_summary.py $OUT > $OUT/summary.log 2>&1
tail -30 $OUT/summary.log
