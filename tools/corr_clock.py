#!/usr/bin/env python
"""shader clock of the correlation launches from a rocprofv3 pass: GRBM_GUI_ACTIVE (busy cycles) of each corr_mfma dispatch over
its duration in the kernel trace.  usage: tools/corr_clock.py <dir of the pass> [skip]"""
import csv, glob, sys, collections
d = sys.argv[1]; skip = int(sys.argv[2]) if len(sys.argv) > 2 else 0
cc = glob.glob(d + "/**/*counter_collection.csv", recursive=True)[0]
kt = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[0]
dur = {}
for r in csv.DictReader(open(kt)):
    dur[r["Dispatch_Id"]] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"]), r["Kernel_Name"])
rows = [r for r in csv.DictReader(open(cc)) if r["Counter_Name"] == "GRBM_GUI_ACTIVE" and "corr_mfma" in r["Kernel_Name"]]
acc = collections.defaultdict(float)
for r in rows:
    acc[r["Dispatch_Id"]] += float(r["Counter_Value"])
ids = sorted(acc, key=int)[skip:]
clk = [acc[i] / dur[i][0] for i in ids if i in dur]      # cycles per ns = GHz (x the number of counter instances summed)
us = [dur[i][0] / 1e3 for i in ids if i in dur]
import numpy as np
print("%d corr_mfma dispatches: duration mean %.1f us (p10 %.1f, p90 %.1f); GRBM_GUI_ACTIVE / ns: mean %.3f (p10 %.3f, p90 %.3f)"
      % (len(clk), np.mean(us), np.percentile(us, 10), np.percentile(us, 90), np.mean(clk), np.percentile(clk, 10), np.percentile(clk, 90)))
