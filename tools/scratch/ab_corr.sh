#!/bin/bash
# A/B: corr schedule off/on -> step time + corr launch time; then FETCH_SIZE with schedule on
export TMPDIR=/tmp
mkdir -p gpurun_out/pmc2
timeout 600 python -m pytest tests -x -q -m gpu 2>&1 | tail -2
for o in 0 1 0 1; do
  RAMP_CORR_ORDER=$o python bench.py --steps 60 --warmup 5 --cpu-steps 0 2>&1 | tail -1 | python -c "
import sys, json; d = json.loads(sys.stdin.read()); print('order=$o', d['value'], d['ms_per_step'], d['roofline'])"
done
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/pmc_$c -o p -- python bench.py --steps 4 --warmup 1 --cpu-steps 0 --prime 64 > /dev/null 2>&1
  f=$(find /tmp/pmc_$c -name '*counter_collection.csv' | head -1)
  python tools/pmc_summary.py $f > gpurun_out/pmc2/$c.txt 2>&1
done
grep -i corr gpurun_out/pmc2/*.txt
