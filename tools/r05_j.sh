#!/bin/bash
export TMPDIR=/tmp
B="--cpu-steps 0 --parity 0 --live-steps 0 --np-steps 0 --inst-steps 0"
for r in 1 2 3; do
for v in "35 x" "0 x" "0 1" "35 1" "15 x"; do set -- $v
  if [ $2 = x ]; then unset RAMP_SELECT_AHEAD; else export RAMP_SELECT_AHEAD=$2; fi
  RAMP_FE_DELAY_US=$1 timeout 600 python bench.py $B 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('delay=$1 ahead=$2', d['value'], 'corr', d['roofline']['mean_launch_us'])"
done; done
