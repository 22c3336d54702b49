#!/bin/bash
# tools/ab.sh "ENV=.. ENV=.." ["ENV=.."...]: bench.py (no parity / cpu legs) once per environment setting, one line each
for e in "$@"; do
  env $e python bench.py --cpu-steps 0 --parity 0 --fp32-leg 0 2>&1 | grep '^{' > /tmp/ab_line.json
  python - "$e" <<'PY'
import json, sys
d = json.load(open("/tmp/ab_line.json")); c = d["config"]
fe = (d.get("roofline_encoder") or {}).get("mean_front_end_us")
print("%-44s %7.1f kf/s  %6.3f ms/step  non-pipelined %6.1f  front end %s us" % (sys.argv[1], d["value"], d["ms_per_step"], c["non_pipelined_kfps"], fe))
PY
done
