#!/bin/bash
# tools/ab_args.sh "ENV=.. ENV=.." ... -- bench args: like ab.sh, with extra bench.py arguments; prints value + the stage legs
envs=()
while [ "$1" != "--" ] && [ $# -gt 0 ]; do envs+=("$1"); shift; done
shift
for e in "${envs[@]}"; do
  env $e python bench.py --cpu-steps 0 --parity 0 --fp32-leg 0 --live-steps 0 "$@" 2>/dev/null | grep '^{' > /tmp/ab_line.json
  python - "$e" <<'PY'
import json, sys
d = json.load(open("/tmp/ab_line.json")); c = d["config"]
u, b, en, r = (d.get(k) or {} for k in ("roofline_update", "roofline_ba", "roofline_encoder", "roofline"))
print("%-36s %7.1f kf/s np %6s | corr %5.1f upd %5.1f (alone %s) ba %5.1f (alone %s) fe %s (tail %s) | E %s..%s live %s" % (
    sys.argv[1], d["value"], c.get("non_pipelined_kfps"), r.get("mean_launch_us", 0), u.get("mean_call_us", 0), u.get("mean_call_us_alone"),
    b.get("mean_call_us", 0), b.get("mean_call_us_alone"), en.get("mean_front_end_us"), en.get("mean_front_end_us_next_to_the_tail"),
    c.get("edges"), c.get("edges_end"), r.get("live_factor_fraction_fine_coarse")))
PY
done
