#!/bin/bash
# fabric traffic of the correlation launch AT THE BENCHMARKED STATE: two PMC passes (one counter each,
# kernel trace only) over the default bench command line, counters collected for corr_mfma launches only
export TMPDIR=/tmp
root=${GRAFT_REPO_ROOT:-$(pwd)}
K=${STEPS:-20}; W=${WARMUP:-5}; L=${LIVE:-40}; I=8
ARGS="--steps $K --warmup $W --cpu-steps 0 --parity 0 --fp32-leg 0 --np-steps 0 --inst-steps $I --live-steps $L --clock-warm-s 0"
mkdir -p gpurun_out/r05pmc
for c in FETCH_SIZE WRITE_SIZE; do
  d=/tmp/pmc_$c; rm -rf $d
  (cd /tmp && timeout 900 rocprofv3 --pmc $c --kernel-trace --kernel-include-regex "corr_mfma" --output-format csv -d $d -o p -- python $root/bench.py $ARGS > /tmp/pmc_$c.log 2>&1)
  echo "$c rc=$?"; tail -2 /tmp/pmc_$c.log | cut -c1-300
  f=$(find $d -name '*counter_collection.csv' | head -1); cp "$f" gpurun_out/r05pmc/$c.csv
done
python tools/pmc_corr_traffic.py gpurun_out/r05pmc/FETCH_SIZE.csv gpurun_out/r05pmc/WRITE_SIZE.csv $K $W $L $I gpurun_out/r05pmc/corr_traffic.json \
  "rocprofv3 --pmc {FETCH_SIZE|WRITE_SIZE} --kernel-trace --kernel-include-regex corr_mfma -- python bench.py $ARGS"
