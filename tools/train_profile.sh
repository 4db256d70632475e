#!/bin/bash
# Development: training-step parity tests, a kernel trace of seven eager steps (per-launch timeline + per-kernel table of one step) and
# the hipGraph step time, on the GPU box.   gpurun -- tools/train_profile.sh [extra bench_train args]
python -m pytest tests/test_train_step.py -x -q -m gpu 2>&1 | tail -3
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/train_prof
rocprofv3 --kernel-trace -d gpurun_out/train_prof -o tr -- python tools/bench_train.py --batch 128 --phonemes 100 --iters 7 "$@" > gpurun_out/train_prof.log 2>&1
grep "train step" gpurun_out/train_prof.log
python tools/train_timeline.py gpurun_out/train_prof/tr_results.db > gpurun_out/train_timeline.txt
python tools/train_timeline.py gpurun_out/train_prof/tr_results.db 3 --by-kernel > gpurun_out/train_by_kernel.md
tail -1 gpurun_out/train_timeline.txt
python tools/bench_train.py --batch 128 --phonemes 100 --iters 20 --graph "$@"
