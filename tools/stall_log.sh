#!/bin/bash
# tools/stall_log.sh: tools/stall_hunt.py with the HIP runtime's own log (AMD_LOG_LEVEL=3); prints the log lines around every gap of
# more than 10 ms between two consecutive log entries inside the steady-state part of the run
export TMPDIR=/tmp
AMD_LOG_LEVEL=${LVL:-3} python tools/stall_hunt.py ${1:-12} 30 plain settle > /tmp/stall_log.out 2> /tmp/stall_log.err
grep "STALL\|cycles with" /tmp/stall_log.out | cut -c1-160
wc -l /tmp/stall_log.err
python - <<'PY'
import re
rx = re.compile(r"\[ts:(\d+)\]|^:(\d):.*?: (\d+) us")
lines = open("/tmp/stall_log.err", errors="replace").read().split("\n")
ts = []
for i, l in enumerate(lines):
    m = re.search(r": (\d{6,}) us", l)
    if m:
        ts.append((int(m.group(1)), i))
print("timestamped lines:", len(ts))
last = None
shown = 0
for k in range(1, len(ts)):
    dt = ts[k][0] - ts[k - 1][0]
    if dt > 10000 and k > len(ts) * 0.4 and shown < 10:
        i0, i1 = ts[k - 1][1], ts[k][1]
        hood = "\n".join(lines[max(0, i0 - 6):i1 + 4])
        if "Using Code Object" in hood or "hipModuleLoad" in hood:
            continue
        shown += 1
        print("==== gap of %.1f ms at log line %d" % (dt / 1e3, i1))
        for l in lines[max(0, i0 - 14):i1 + 6]:
            print("   ", l[:220])
PY
