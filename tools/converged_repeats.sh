#!/bin/bash
# tools/converged_repeats.sh [N]: N runs of bench.py's legs on this box; one line per run with every leg's rate and LegStats
# (a leg that fell back to host-driven frames, or contains a stall, shows in host_frames / settles / max step) -- VERDICT r5 #6
N=${1:-5}
for r in $(seq 1 $N); do
  python bench.py --cpu-steps 0 --parity 0 --fp32-leg 0 --fp32-leg 0 --steps 100 2>/dev/null | grep "^{" | python -c "
import sys, json
d = json.loads(sys.stdin.read()); c = d['config']; L = c['legs']
f = lambda k: '%s p50/p90/max %s dev/host %d/%d settles %d throttle %.0f ms' % (k, L[k]['host_step_ms_p50_p90_max'], L[k]['device_frames'], L[k]['host_frames'], L[k]['settles'], L[k]['throttle_ms'])
print('run $r: timed %.1f np %.1f own %.1f converged %.1f kf/s | %s | %s' % (d['value'], c['non_pipelined_kfps'], c['own_stream_kfps'], c.get('converged_kfps') or 0, f('converged'), f('live')))
"
done
