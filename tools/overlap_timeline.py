#!/usr/bin/env python3
"""Development: start / end of every kernel of the last few steps of a rocprofv3 --kernel-trace csv, relative to the first listed
kernel, with the queue it ran on -- to see what overlaps in the two-stream launch mode.  python tools/overlap_timeline.py <dir> [n]"""
import csv, glob, sys, re
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 16
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
rows = [r for r in rows if "esmi" in r["Kernel_Name"]][-n:]
t0 = int(rows[0]["Start_Timestamp"])
for r in rows:
    name = re.sub(r"\(.*$", "", re.sub(r"^void ", "", r["Kernel_Name"])).replace("esmi::", "")[:44]
    print(f"{(int(r['Start_Timestamp']) - t0) / 1e3:9.1f} -> {(int(r['End_Timestamp']) - t0) / 1e3:9.1f} us  q{r.get('Queue_Id', '?'):>3}  {name}")
