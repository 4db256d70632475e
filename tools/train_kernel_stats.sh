#!/bin/bash
# On the GPU box: kernel trace of the eager training step -> gpurun_out/<tag>/train_kernel_stats.md
# usage: tools/train_kernel_stats.sh <tag> [bench_train args]
set -u
TAG=${1:-r03}; shift || true
REPO=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$REPO/gpurun_out/$TAG; mkdir -p $OUT
ARGS="--batch 128 --phonemes 100 --iters 7 $*"
python $REPO/tools/bench_train.py $ARGS 2>/dev/null | tail -1 > $OUT/train_unprofiled.txt
python $REPO/tools/bench_train.py $ARGS --graph 2>/dev/null | tail -1 >> $OUT/train_unprofiled.txt
(cd /tmp && TMPDIR=/tmp timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/train_stats -o r -- python $REPO/tools/bench_train.py $ARGS > $OUT/train_stats.log 2>&1 < /dev/null)
python - $OUT "$ARGS" <<'PY'
import csv, glob, sys, collections, re
out, args = sys.argv[1], sys.argv[2]
f = glob.glob(out + "/train_stats/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
def short(r):
    n = re.sub(r"^void ", "", r["Kernel_Name"]); n = re.sub(r"\(.*$", "", n); return n.replace("esmi::", "")[:72]
# one optimizer launch per step delimits the steps; the steady state = the steps after the 3 warm-up ones (set-up copies, first-use packing excluded)
ends = [i for i, r in enumerate(rows) if "train_adamw" in r["Kernel_Name"]]
win = rows[ends[3] + 1: ends[-1] + 1]
steps = len(ends) - 4
agg = collections.defaultdict(lambda: [0, 0.0])
for r in win:
    a = agg[short(r)]; a[0] += 1; a[1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
tot = sum(a[1] for a in agg.values()); nl = sum(a[0] for a in agg.values())
with open(out + "/train_kernel_stats.md", "w") as o:
    o.write(f"`rocprofv3 --kernel-trace --stats -- python tools/bench_train.py {args}`; the last {steps} of {len(ends)} eager steps (from one optimizer launch to the next)\n\n")
    o.write("un-profiled: " + " | ".join(l.strip() for l in open(out + "/train_unprofiled.txt")) + "\n\n")
    o.write(f"**{nl / steps:.0f} launches** and {tot / steps / 1e3:.2f} ms of kernel time per step\n\n| kernel | launches / step | average | ms / step |\n|---|---:|---:|---:|\n")
    for n, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:45]:
        o.write(f"| `{n}` | {c / steps:.1f} | {t / c:.1f} us | {t / steps / 1e3:.3f} |\n")
print(open(out + "/train_kernel_stats.md").read()[:4000])
PY
