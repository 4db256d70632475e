#!/bin/bash
# On the GPU box: kernel trace of the HiFi-GAN generator forward -> gpurun_out/<tag>/vocoder_kernel_stats.md
set -u
TAG=${1:-r03}; REPO=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$REPO/gpurun_out/$TAG; mkdir -p $OUT
python $REPO/tools/bench_vocoder.py --batch 16 --iters 20 2>/dev/null | tail -2 > $OUT/vocoder_unprofiled.txt
(cd /tmp && TMPDIR=/tmp timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/voc_stats -o r -- python $REPO/tools/bench_vocoder.py --batch 16 --iters 20 > $OUT/voc_stats.log 2>&1 < /dev/null)
python - $OUT <<'PY'
import csv, glob, sys, collections, re
out = sys.argv[1]
f = glob.glob(out + "/voc_stats/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
rows = [r for r in rows if "convgemm" in r["Kernel_Name"] or "resblock" in r["Kernel_Name"] or "conv_to1" in r["Kernel_Name"]]
ends = [i for i, r in enumerate(rows) if "conv_to1" in r["Kernel_Name"]]          # conv_post closes a forward
win = rows[ends[-11] + 1: ends[-1] + 1]; n = 10
agg = collections.defaultdict(lambda: [0, 0.0])
for r in win:
    k = re.sub(r"^void ", "", r["Kernel_Name"]); k = re.sub(r"\(.*$", "", k).replace("esmi::", "")
    a = agg[k]; a[0] += 1; a[1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
tot = sum(a[1] for a in agg.values())
with open(out + "/vocoder_kernel_stats.md", "w") as o:
    o.write("`rocprofv3 --kernel-trace --stats -- python tools/bench_vocoder.py --batch 16 --iters 20`; the last 10 forwards\n\nun-profiled: " + " | ".join(l.strip() for l in open(out + "/vocoder_unprofiled.txt")) + f"\n\n{sum(a[0] for a in agg.values()) / n:.0f} launches and {tot / n:.0f} us of kernel time per forward\n\n| kernel | launches / forward | average | share |\n|---|---:|---:|---:|\n")
    for k, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        o.write(f"| `{k}` | {c / n:.0f} | {t / c:.1f} us | {100 * t / tot:.1f} % |\n")
print(open(out + "/vocoder_kernel_stats.md").read())
PY
