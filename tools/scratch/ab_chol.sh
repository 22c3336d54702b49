#!/bin/bash
export TMPDIR=/tmp
for lib in build_variants/*.so; do
  rm -rf /tmp/pg; RAMP_HIP_LIB=$PWD/$lib timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pg -o r -- python bench.py --steps 20 --warmup 3 --cpu-steps 0 > /tmp/pg.log 2>&1
  echo "$lib $(grep ba_chol /tmp/pg/r_kernel_stats.csv | cut -d, -f1-4 | cut -c1-20,60-120) $(grep metric /tmp/pg.log | cut -c80-130)"
done
