#!/bin/bash
# tools/stall_malloc.sh: every hipMalloc / hipFree of a tools/stall_hunt.py run (AMD_LOG_LEVEL=3) with the kernels launched around it
export TMPDIR=/tmp
v=${1:-settle}
AMD_LOG_LEVEL=3 python tools/stall_hunt.py 8 30 plain $v > /tmp/sm.out 2> /tmp/sm.err
grep "cycles with\|STALL" /tmp/sm.out | cut -c1-150
python - <<'PY'
import re
lines = open("/tmp/sm.err", errors="replace").read().split("\n")
n = len(lines)
last_k = []
out = 0
for i, l in enumerate(lines):
    if "ShaderName" in l:
        last_k.append(re.sub(r".*ShaderName : ", "", l)[:60]); last_k = last_k[-2:]
    if re.search(r"\b(hipMalloc|hipFree|hipHostMalloc|hipExtMallocWithFlags|hipHostFree) \(", l) and "Returned" not in l:
        ts = re.search(r": (\d{6,}) us", l)
        nxt = ""
        for j in range(i + 1, min(n, i + 400)):
            if "ShaderName" in lines[j]:
                nxt = re.sub(r".*ShaderName : ", "", lines[j])[:60]; break
        ret = ""
        for j in range(i + 1, min(n, i + 60)):
            if "Returned" in lines[j] and ("hipMalloc" in lines[j] or "hipFree" in lines[j] or "Malloc" in lines[j]):
                ret = re.sub(r".*Returned", "", lines[j])[:70]; break
        print("line %7d/%d t=%s %s | after %s | before %s | %s" % (i, n, ts.group(1) if ts else "?", re.sub(r"^.*?(hip\w+ \([^)]*\)).*$", r"\1", l)[:70], last_k[-1:] , nxt, ret))
        out += 1
print("allocation calls:", out)
PY
