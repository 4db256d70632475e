#!/usr/bin/env python3
"""Development: per-dispatch durations of the last HiFi-GAN forward in a rocprofv3 --kernel-trace csv.
python tools/voc_trace_summary.py <kernel_trace.csv> <launches per forward>"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
n = int(sys.argv[2])
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
rows = [r for r in rows if "convgemm" in r["Kernel_Name"] or "resblock" in r["Kernel_Name"]][-n:]
t0 = int(rows[0]["Start_Timestamp"])
tot = 0
for i, r in enumerate(rows):
    d = int(r["End_Timestamp"]) - int(r["Start_Timestamp"]); tot += d
    name = r["Kernel_Name"].split("(")[0][-40:]
    print(f"{i:3d} {(int(r['Start_Timestamp'])-t0)/1e3:9.1f}us  {d/1e3:8.1f}us  grid={r.get('Grid_Size_X','?')} wg={r.get('Workgroup_Size_X','?')} lds={r.get('LDS_Block_Size','?')} {name}")
print("sum of kernel time %.1f us, span %.1f us" % (tot / 1e3, (int(rows[-1]["End_Timestamp"]) - t0) / 1e3))
