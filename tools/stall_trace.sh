#!/bin/bash
# tools/stall_trace.sh: tools/stall_hunt.py under rocprofv3 (kernel trace + HIP runtime API trace, no counters): which API call
# blocks, and between which kernels the GPU idles, when a frame behind a state_dict() hand-back takes ~50 ms
export TMPDIR=/tmp
root=${GRAFT_REPO_ROOT:-$(pwd)}
d=/tmp/stall_trace; rm -rf $d
(cd /tmp && timeout 900 rocprofv3 --kernel-trace --hip-runtime-trace ${TRACE_EXTRA:-} --output-format csv -d $d -o t -- python $root/tools/stall_hunt.py ${1:-20} 30 plain ${2:-1} > /tmp/stall_trace.log 2>&1)
grep "STALL\|cycles with" /tmp/stall_trace.log | cut -c1-300
python - $d <<'PY'
import csv, glob, sys, os
d = sys.argv[1]
api = glob.glob(os.path.join(d, "**", "*hip_api_trace.csv"), recursive=True)
ker = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)
print("files:", api, ker)
if api:
    rows = list(csv.DictReader(open(api[0])))
    slow = [r for r in rows if (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) > 10e6]
    t0 = int(rows[0]["Start_Timestamp"])
    print("HIP API calls longer than 10 ms: %d of %d" % (len(slow), len(rows)))
    for r in slow[-40:]:
        print("  %-40s %8.2f ms at t=%.3f s tid=%s" % (r["Function"], (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6, (int(r["Start_Timestamp"]) - t0) / 1e9, r.get("Thread_Id")))
for hs in glob.glob(os.path.join(d, "**", "*hsa_api_trace.csv"), recursive=True):
    rows = list(csv.DictReader(open(hs)))
    t0 = int(rows[0]["Start_Timestamp"])
    slow = [r for r in rows if (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) > 5e6]
    print("HSA API calls longer than 5 ms: %d of %d" % (len(slow), len(rows)))
    for r in slow[-60:]:
        print("  %-46s %8.2f ms at t=%.3f s tid=%s" % (r["Function"], (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6, (int(r["Start_Timestamp"]) - t0) / 1e9, r.get("Thread_Id")))
if ker:
    rows = sorted(csv.DictReader(open(ker[0])), key=lambda r: int(r["Start_Timestamp"]))
    t0 = int(rows[0]["Start_Timestamp"])
    end = 0
    gaps = []
    for i, r in enumerate(rows):
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        if end and s - end > 10e6:
            gaps.append((s - end, i))
        end = max(end, e)
    print("GPU idle gaps longer than 10 ms: %d" % len(gaps))
    for g, i in gaps[-30:]:
        prev = rows[i - 1]; nxt = rows[i]
        print("  %.2f ms idle at t=%.3f s: after %s (queue %s) before %s (queue %s)" % (g / 1e6, (int(nxt["Start_Timestamp"]) - t0) / 1e9, prev["Kernel_Name"][:50], prev.get("Queue_Id"), nxt["Kernel_Name"][:50], nxt.get("Queue_Id")))
    # the neighbourhood of the gaps that fall between two tracker kernels (not the hand-back's own copies)
    apirows = list(csv.DictReader(open(api[0]))) if api else []
    shown = 0
    for g, i in reversed(gaps):
        names = rows[i - 1]["Kernel_Name"] + rows[i]["Kernel_Name"]
        around = " ".join(r["Kernel_Name"] for r in rows[max(0, i - 40):i + 40])
        if "rocclr" in names or "at::native" in names or "Cijk" in names or shown >= 4 or "upd_gru_kernel" not in around:
            continue
        shown += 1
        gs, ge = int(rows[i - 1]["End_Timestamp"]), int(rows[i]["Start_Timestamp"])
        print("---- gap of %.2f ms: kernels around it (time relative to the gap's start, us; queue; duration us)" % (g / 1e6))
        for r in rows[max(0, i - 14):i + 10]:
            print("   %+10.1f  q%s  %7.1f  %s" % ((int(r["Start_Timestamp"]) - gs) / 1e3, r.get("Queue_Id"), (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, r["Kernel_Name"][:70]))
        print("     HIP API calls overlapping the gap (start rel. us, duration us):")
        for a in apirows:
            a0, a1 = int(a["Start_Timestamp"]), int(a["End_Timestamp"])
            if a1 > gs and a0 < ge and (a1 - a0) > 200e3:
                print("       %+10.1f %9.1f %s" % ((a0 - gs) / 1e3, (a1 - a0) / 1e3, a["Function"]))
    long = [r for r in rows if int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) > 10e6]
    print("kernels longer than 10 ms: %d" % len(long))
    for r in long[-20:]:
        print("  %-50s %.2f ms at t=%.3f s queue %s" % (r["Kernel_Name"][:50], (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6, (int(r["Start_Timestamp"]) - t0) / 1e9, r.get("Queue_Id")))
PY
