#!/bin/bash
export TMPDIR=/tmp
for lib in build_variants/*.so; do
  rm -rf /tmp/pg; RAMP_HIP_LIB=$PWD/$lib timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pg -o r -- python bench.py --steps 20 --warmup 3 --cpu-steps 0 > /dev/null 2>&1
  echo "$lib $(grep "upd_nbr\|upd_gru" /tmp/pg/r_kernel_stats.csv | cut -d, -f2-4)"
done
