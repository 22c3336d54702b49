#!/bin/bash
# A/B kernel builds: for each build_variants/*.so run the bench and print step time + corr launch time
export TMPDIR=/tmp
[ -n "$RUN_TESTS" ] && timeout 900 python -m pytest tests -x -q -m gpu 2>&1 | tail -3
for lib in build_variants/*.so; do
  RAMP_HIP_LIB=$PWD/$lib python bench.py --steps ${STEPS:-30} --warmup 5 --cpu-steps 0 2>&1 | tail -1 | python -c "
import sys, json; d = json.loads(sys.stdin.read()); r = d['roofline']; print('$lib', d['value'], d['ms_per_step'], r['mean_launch_us'], r['frac'])"
done
