#!/usr/bin/env python
"""one steady-state frame of a rocprofv3 kernel trace as a timeline: start offset, duration, stream/queue, name"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "motionmag" in r["Kernel_Name"]]
back = int(sys.argv[2]) if len(sys.argv) > 2 else 3
s, t = idx[-back - 1], idx[-back]
t0 = int(rows[s]["Start_Timestamp"])
keys = [k for k in ("Stream_Id", "Queue_Id") if k in rows[0]]
for r in rows[s:t + 1]:
    a, b = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print("%8.1f %7.1f  %s  %s" % ((a - t0) / 1e3, (b - a) / 1e3, "/".join(r[k] for k in keys), r["Kernel_Name"][:70]))
