#!/usr/bin/env python
"""GPU busy time per steady-state frame from a rocprofv3 --kernel-trace CSV (union of kernel intervals over all
streams, frames delimited by the keyframe motion-test kernel), plus the per-kernel totals of those frames.
usage: trace_busy.py kernel_trace.csv [n_frames]"""
import collections
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
nf = int(sys.argv[2]) if len(sys.argv) > 2 else 10
idx = [i for i, r in enumerate(rows) if "motionmag" in r["Kernel_Name"]]
s, t = idx[-nf - 1], idx[-1]
iv = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in rows[s:t])
busy, (cs, ce) = 0, iv[0]
for a, b in iv[1:]:
    if a > ce:
        busy += ce - cs
        cs, ce = a, b
    else:
        ce = max(ce, b)
busy += ce - cs
span = int(rows[t]["Start_Timestamp"]) - int(rows[s]["Start_Timestamp"])
print("%d frames: span %.1f us/frame (traced run), GPU busy (union) %.1f us/frame, kernels %.1f/frame"
      % (nf, span / 1e3 / nf, busy / 1e3 / nf, (t - s) / nf))
acc = collections.Counter()
cnt = collections.Counter()
for r in rows[s:t]:
    k = r["Kernel_Name"][:60]
    acc[k] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    cnt[k] += 1
for k, v in acc.most_common(int(sys.argv[3]) if len(sys.argv) > 3 else 40):
    print("%8.1f us/frame %5.1f x  %s" % (v / 1e3 / nf, cnt[k] / nf, k))
