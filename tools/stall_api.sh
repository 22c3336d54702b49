#!/bin/bash
# tools/stall_api.sh: which HIP API calls does a hand-back (settle() + re-entry) make that steady tracking does not?  Two runs of
# tools/stall_hunt.py under AMD_LOG_LEVEL=3 (hand-back every cycle / never), histogram of the API names in the cycled part of each
export TMPDIR=/tmp
for v in settle 0; do
  AMD_LOG_LEVEL=3 python tools/stall_hunt.py 8 30 plain $v > /tmp/stall_api_$v.out 2> /tmp/stall_api_$v.err
  grep "cycles with" /tmp/stall_api_$v.out
done
python - <<'PY'
import re, collections
def hist(path):
    lines = open(path, errors="replace").read().split("\n")
    lines = lines[int(len(lines) * 0.5):]
    h = collections.Counter()
    for l in lines:
        m = re.search(r"\b(hip[A-Za-z_]+) \(", l)
        if m and "Returned" not in l:
            h[m.group(1)] += 1
        m = re.search(r"(Allocat\w+|hsa_amd_memory_\w+|memory_lock|SVM|pin\w*|unpin\w*|Free\w*|release\w*)", l)
        if m and "hip" not in l[:60]:
            h["[rt] " + m.group(1)] += 1
    return h
a, b = hist("/tmp/stall_api_settle.err"), hist("/tmp/stall_api_0.err")
print("%-40s %10s %10s" % ("call", "hand-backs", "steady"))
for k in sorted(set(a) | set(b), key=lambda k: -(a[k] + b[k])):
    print("%-40s %10d %10d" % (k, a[k], b[k]))
# the runtime's own lines (not API entry/exit) that only the hand-back run has
def rt(path):
    lines = open(path, errors="replace").read().split("\n")
    lines = lines[int(len(lines) * 0.5):]
    h = collections.Counter()
    for l in lines:
        m = re.match(r":\d:([\w\.]+)\s*:\d+\s*: \d+ us:\s*(\[[^\]]*\])?\s*(.{0,60})", l)
        if m and "hip" not in m.group(3)[:4]:
            h[m.group(1) + " | " + re.sub(r"0x[0-9a-f]+|\d+", "#", m.group(3))[:50]] += 1
    return h
a, b = rt("/tmp/stall_api_settle.err"), rt("/tmp/stall_api_0.err")
print("\nruntime log lines by source file / text, seen only (or 2x more) with hand-backs:")
for k in sorted(a, key=lambda k: -a[k]):
    if a[k] > 2 * b.get(k, 0) and a[k] >= 4:
        print("%8d %8d  %s" % (a[k], b.get(k, 0), k))
PY
