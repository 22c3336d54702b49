#!/usr/bin/env python
"""tools/timeline_gaps.py KERNEL_TRACE_CSV [N_TAIL_FRAMES]: per-queue busy / idle accounting of the steady-state tail of a
rocprofv3 --kernel-trace run of bench.py.  A frame is delimited by consecutive corr_mfma launches (one per update())."""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
ntail = int(sys.argv[2]) if len(sys.argv) > 2 else 60
for r in rows:
    r["s"], r["e"] = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
rows.sort(key=lambda r: r["s"])
corr = [r for r in rows if "corr_mfma" in r["Kernel_Name"]]
if len(corr) <= ntail + 2:
    print(len(rows), "rows;", len(corr), "corr launches;", collections.Counter(r["Kernel_Name"][:40] for r in rows).most_common(8)); sys.exit(1)
t0, t1 = corr[-ntail - 1]["s"], corr[-1]["s"]
main_q = corr[-1]["Queue_Id"]
win = [r for r in rows if t0 <= r["s"] < t1]
print("window: %d frames, %.1f us per frame; main queue %s" % (ntail, (t1 - t0) / ntail / 1e3, main_q))
byq = collections.defaultdict(list)
for r in win:
    byq[r["Queue_Id"]].append(r)
for q, rs in sorted(byq.items(), key=lambda kv: -len(kv[1])):
    busy = sum(r["e"] - r["s"] for r in rs)
    # union of intervals (kernels on one queue can overlap when independent)
    cur_s, cur_e, union = None, None, 0
    for r in rs:
        if cur_e is None or r["s"] > cur_e:
            if cur_e is not None:
                union += cur_e - cur_s
            cur_s, cur_e = r["s"], r["e"]
        else:
            cur_e = max(cur_e, r["e"])
    union += cur_e - cur_s
    print("queue %s: %5.1f launches/frame, kernel time %.1f us/frame, busy (union) %.1f us/frame, idle %.1f us/frame"
          % (q, len(rs) / ntail, busy / ntail / 1e3, union / ntail / 1e3, ((t1 - t0) - union) / ntail / 1e3))
# the main queue in order: kernels + the gap before each, averaged per kernel name (position-agnostic)
rs = byq[main_q]
gap_by = collections.defaultdict(lambda: [0, 0, 0])
prev_e = None
for r in rs:
    g = 0 if prev_e is None else max(0, r["s"] - prev_e)
    prev_e = max(prev_e or 0, r["e"])
    k = r["Kernel_Name"][:60]
    gap_by[k][0] += 1; gap_by[k][1] += r["e"] - r["s"]; gap_by[k][2] += g
print("main queue, per frame: kernel | launches | kernel us | idle-before us")
for k, (n, d, g) in sorted(gap_by.items(), key=lambda kv: -(kv[1][1] + kv[1][2])):
    print("  %-60s %5.1f %8.1f %8.1f" % (k, n / ntail, d / ntail / 1e3, g / ntail / 1e3))
# one steady-state frame, every queue, in start order
fr = [r for r in rows if corr[-3]["s"] - 400_000 <= r["s"] < corr[-2]["s"] - 400_000]
base = fr[0]["s"]
print("one frame (all queues): t_start us | dur us | queue/stream | kernel")
for r in fr:
    print("  %8.1f %7.1f  q%s/s%s  %s" % ((r["s"] - base) / 1e3, (r["e"] - r["s"]) / 1e3, r["Queue_Id"], r["Stream_Id"], r["Kernel_Name"][:70]))
