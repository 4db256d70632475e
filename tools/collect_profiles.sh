#!/bin/bash
# Run on the GPU box (via gpurun): un-profiled bench line, kernel-trace stats pass, one PMC pass per counter group.
# usage: tools/collect_profiles.sh <tag> [bench args...]   -> gpurun_out/<tag>/{bench_unprofiled.json,stats,pmc_*}/ + summaries
# Every step is wrapped in `timeout`: a hung profiler must not eat the GPU budget.
set -u
TAG=${1:-r02}; shift || true
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 300 python $REPO/bench.py "$@" > $OUT/bench_unprofiled.json 2> $OUT/bench_unprofiled.err < /dev/null
PROF_ARGS="--no-cpu-baseline --no-extras --no-auto-launch"
timeout 240 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o r -- \
    python $REPO/bench.py --steps 10 --warmup 3 --event-every 1 $PROF_ARGS "$@" > $OUT/stats.log 2>&1 < /dev/null
if [ "${PMC:-1}" == "1" ]; then
for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE" \
           "SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_MFMA SQ_INSTS_VALU SQ_INSTS_LDS" "SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY" \
           "SQ_ACTIVE_INST_VALU SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_COEXEC_CYCLES SQ_ACTIVE_INST_ANY"; do
  name=pmc_$(echo $grp | cut -d' ' -f1)
  timeout 240 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $OUT/$name -o r -- \
      python $REPO/bench.py --steps 3 --warmup 2 $PROF_ARGS "$@" > $OUT/$name.log 2>&1 < /dev/null
done
fi
cd $REPO
timeout 120 python tools/prof_summary.py $OUT > $OUT/summary.log 2>&1 < /dev/null
timeout 60 python tools/step_timeline.py $OUT/stats > $OUT/step_timeline.txt 2>&1 < /dev/null
tail -30 $OUT/summary.log
