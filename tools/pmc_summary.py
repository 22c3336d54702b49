#!/usr/bin/env python
"""Aggregate a rocprofv3 --pmc counter_collection CSV per kernel: mean / max counter value per dispatch, and the mean
over the steady-state dispatches (value >= 0.8 max: the launches at the full sliding-window size).
usage: pmc_summary.py counter_collection.csv [name-filter ...]"""
import csv, sys, collections
rows = csv.DictReader(open(sys.argv[1]))
filt = sys.argv[2:]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    name = r.get("Kernel_Name") or r.get("Kernel Name") or ""
    if filt and not any(f in name for f in filt):
        continue
    acc[name[:70]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for name, cs in acc.items():
    for c, v in cs.items():
        mx = max(v)
        st = [x for x in v if x >= 0.8 * mx] or v
        print("%-72s %-28s n=%-6d mean=%.6g max=%.6g steady(n=%d)=%.6g" % (name, c, len(v), sum(v) / len(v), mx, len(st), sum(st) / len(st)))
