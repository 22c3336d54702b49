#!/bin/bash
# rocprofv3 PMC passes (one counter set per run, no tracing domains besides the kernel trace) over a short
# bench.py run; prints per-kernel mean/max per dispatch for kernels matching $FILTER.
# usage: tools/pmc_passes.sh OUTDIR "CTR_A CTR_B ..." ["CTR_C ..." ...]
export TMPDIR=/tmp
out=$1; shift
mkdir -p $out
i=0
for set in "$@"; do
  i=$((i+1))
  d=/tmp/pmcpass_$i
  rm -rf $d
  timeout 600 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $d -o p -- python bench.py --steps ${STEPS:-4} --warmup 1 --cpu-steps 0 --prime ${PRIME:-64} > /tmp/pmcpass_$i.log 2>&1
  f=$(find $d -name '*counter_collection.csv' | head -1)
  if [ -z "$f" ]; then echo "pass $i ($set): no output"; tail -3 /tmp/pmcpass_$i.log; continue; fi
  python tools/pmc_summary.py $f ${FILTER:-} > $out/pass_$i.txt 2>&1
done
cat $out/pass_*.txt
