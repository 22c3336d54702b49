#!/usr/bin/env python
"""Trim a rocprofv3 --kernel-trace --stats CSV (…_kernel_stats.csv) to a readable summary.
usage: summarize_rocprof.py in.csv out.md [title]"""
import csv
import sys

src, dst = sys.argv[1], sys.argv[2]
title = sys.argv[3] if len(sys.argv) > 3 else src
rows = list(csv.DictReader(open(src)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
calls = sum(int(r["Calls"]) for r in rows)
with open(dst, "w") as f:
    f.write("# %s\n\nrocprofv3 --kernel-trace --stats; %d kernels, %d launches, %.1f ms total GPU kernel time\n\n" %
            (title, len(rows), calls, tot / 1e6))
    f.write("| kernel | calls | total ms | avg us | % |\n|---|---:|---:|---:|---:|\n")
    for r in rows[:45]:
        name = r["Name"].replace("|", "/")
        if len(name) > 96:
            name = name[:93] + "..."
        f.write("| `%s` | %s | %.2f | %.1f | %.2f |\n" % (name, r["Calls"], float(r["TotalDurationNs"]) / 1e6,
                                                       float(r["AverageNs"]) / 1e3, float(r["Percentage"])))
print("wrote", dst)
