#!/usr/bin/env python3
"""Per-kernel timeline (gap before + duration) of the last forward step in a rocprofv3 kernel trace csv.
usage: step_timeline.py <dir containing *kernel_trace.csv>"""
import csv, glob, sys
path = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(path)), key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "mel_decoder" in r["Kernel_Name"]]
# a step from the middle of the timed region (3 warm-up + 10 timed steps in tools/collect_profiles.sh): the trace's LAST decoder launches
# belong to bench.py's legs behind the timed region (clock probe: a device->host copy and a sync per launch)
k = 9 if len(idx) > 10 else len(idx) - 1
a, b = idx[k - 1], idx[k]
prev_end = int(rows[a]["End_Timestamp"])
tot_gap = tot_dur = 0.0
for r in rows[a + 1:b + 1]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = r["Kernel_Name"].replace("void esmi::", "").split("(")[0][:48]
    print("%-48s gap %7.2f us  dur %8.2f us" % (name, (s - prev_end) / 1e3, (e - s) / 1e3))
    tot_gap += (s - prev_end) / 1e3; tot_dur += (e - s) / 1e3
    prev_end = e
print("sum of gaps %.2f us, sum of durations %.2f us" % (tot_gap, tot_dur))
