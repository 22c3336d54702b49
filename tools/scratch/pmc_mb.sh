#!/bin/bash
# PMC passes over a micro-benchmark script: pmc_mb.sh SCRIPT OUT "set1" "set2" ...
export TMPDIR=/tmp
script=$1; out=$2; shift; shift
mkdir -p $out
i=0
for set in "$@"; do
  i=$((i+1)); d=/tmp/pmcmb_$i; rm -rf $d
  timeout 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $d -o p -- python $script > /tmp/pmcmb_$i.log 2>&1
  f=$(find $d -name '*counter_collection.csv' | head -1)
  if [ -z "$f" ]; then echo "pass $i ($set): no output"; tail -3 /tmp/pmcmb_$i.log; continue; fi
  FILTER=${FILTER:-upd_} python tools/pmc_summary.py $f ${FILTER:-upd_} > $out/pass_$i.txt 2>&1
done
cat $out/pass_*.txt
