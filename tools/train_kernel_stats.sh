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
rows = list(csv.DictReader(open(f)))
steps = 10   # 3 warm-up + 7 timed (tools/bench_train.py)
agg = collections.defaultdict(lambda: [0, 0.0])
for r in rows:
    n = re.sub(r"^void ", "", r["Kernel_Name"]); n = re.sub(r"\(.*$", "", n); n = n.replace("esmi::", "")
    a = agg[n[:72]]; a[0] += 1; a[1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
tot = sum(a[1] for a in agg.values()); nl = sum(a[0] for a in agg.values())
with open(out + "/train_kernel_stats.md", "w") as o:
    o.write(f"`rocprofv3 --kernel-trace --stats -- python tools/bench_train.py {args}` ({steps} eager steps incl. 3 warm-up; first-step packing included)\n\n")
    o.write("un-profiled: " + " | ".join(l.strip() for l in open(out + "/train_unprofiled.txt")) + "\n\n")
    o.write(f"{nl / steps:.0f} launches and {tot / steps / 1e3:.2f} ms of kernel time per step\n\n| kernel | launches / step | average | ms / step |\n|---|---:|---:|---:|\n")
    for n, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:40]:
        o.write(f"| `{n}` | {c / steps:.1f} | {t / c:.1f} us | {t / steps / 1e3:.3f} |\n")
print(open(out + "/train_kernel_stats.md").read()[:3500])
PY
