#!/bin/bash
# rocprofv3 PMC passes (ONE counter set per run, no tracing domains besides the kernel trace) over a short bench.py
# run; writes per-kernel mean/max per dispatch for kernels matching $FILTER into OUTDIR/pass_<i>.txt.
# usage: tools/pmc_passes.sh OUTDIR "CTR_A CTR_B ..." ["CTR_C ..." ...]
export TMPDIR=/tmp
out=$1; shift
mkdir -p $out
root=${GRAFT_REPO_ROOT:-$(pwd)}
i=0
for set in "$@"; do
  i=$((i+1))
  d=/tmp/pmcpass_$i
  rm -rf $d
  (cd /tmp && timeout 600 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $d -o p -- python $root/bench.py --steps ${STEPS:-6} --warmup 2 --cpu-steps 0 --parity 0 --fp32-leg 0 --np-steps 0 --inst-steps 0 --clock-warm-s 0 --live-steps 0 --prime ${PRIME:-64} ${BENCH_ARGS:-} > /tmp/pmcpass_$i.log 2>&1)
  f=$(find $d -name '*counter_collection.csv' | head -1)
  if [ -z "$f" ]; then echo "pass $i ($set): no output"; tail -3 /tmp/pmcpass_$i.log; continue; fi
  echo "# rocprofv3 --pmc $set --kernel-trace -- python bench.py --steps ${STEPS:-6} --warmup 2 --cpu-steps 0 --parity 0 --fp32-leg 0 --np-steps 0 --inst-steps 0 --clock-warm-s 0 --live-steps 0 --prime ${PRIME:-64} ${BENCH_ARGS:-}" > $out/pass_$i.txt
  python tools/pmc_summary.py $f ${FILTER:-} >> $out/pass_$i.txt 2>&1
done
cat $out/pass_*.txt
