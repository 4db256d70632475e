#!/usr/bin/env python3
"""Per-launch timeline of one training step in a rocprofv3 result database (rocprofv3 --kernel-trace -d DIR -o NAME: DIR/NAME_results.db;
a step = from one optimizer launch to the next).
usage: train_timeline.py <results.db> [step index from the end, default 3] [--by-kernel]"""
import sqlite3, sys
from collections import defaultdict
db = sqlite3.connect(sys.argv[1])
cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
rows = [dict(zip(cols, r)) for r in db.execute("select * from kernels order by start")]
idx = [i for i, r in enumerate(rows) if "train_adamw" in r["name"]]
k = int(sys.argv[2]) if len(sys.argv) > 2 and sys.argv[2].isdigit() else 3
a, b = idx[-k], idx[-k + 1]
short = lambda r: r["name"].replace("void ", "").replace("esmi::", "").split("(")[0][:60]
if "--by-kernel" in sys.argv:
    agg = defaultdict(lambda: [0, 0.0])
    for r in rows[a + 1:b + 1]:
        agg[short(r)][0] += 1; agg[short(r)][1] += (r["end"] - r["start"]) / 1e3
    print("| kernel | launches / step | us / step |\n|---|---:|---:|")
    for n, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print("| `%s` | %d | %.1f |" % (n, c, t))
    print("| **total** | **%d** | **%.1f** |" % (b - a, sum(t for _, t in agg.values())))
    sys.exit(0)
prev = rows[a]["end"]; tg = td = 0.0
for n, r in enumerate(rows[a + 1:b + 1]):
    print("%3d %-60s gap %7.2f dur %8.2f grid %sx%sx%s wg %s" % (n + 1, short(r), (r["start"] - prev) / 1e3, (r["end"] - r["start"]) / 1e3,
          r.get("grid_x"), r.get("grid_y"), r.get("grid_z"), r.get("workgroup_x")))
    tg += (r["start"] - prev) / 1e3; td += (r["end"] - r["start"]) / 1e3; prev = r["end"]
print("%d launches, sum of gaps %.1f us, sum of durations %.1f us" % (b - a, tg, td))
