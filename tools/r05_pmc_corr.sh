#!/bin/bash
# fabric traffic of the correlation launch AT THE BENCHMARKED STATE: two PMC passes (one counter each,
# kernel trace only) over the default bench command line, counters collected for corr_mfma launches only
export TMPDIR=/tmp
root=${GRAFT_REPO_ROOT:-$(pwd)}
K=${STEPS:-20}; W=${WARMUP:-5}; L=${LIVE:-40}; I=8
ARGS="--steps $K --warmup $W --cpu-steps 0 --parity 0 --fp32-leg 0 --np-steps 0 --inst-steps $I --live-steps $L --clock-warm-s 0"
# MIXED=0: the fp32 tracker (corr_mfma_kernel<CorrX2>, rows of 896 floats) -> gpurun_out/r05pmc/corr_traffic_fp32.json
SUF=""; ROWB=1792; KNAME="corr_mfma_kernel<_Float16, true>"
if [ "${MIXED:-1}" = "0" ]; then ARGS="$ARGS --mixed 0"; SUF="_fp32"; ROWB=3584; KNAME="corr_mfma_kernel<CorrX2, true>"; fi
mkdir -p gpurun_out/r05pmc
for c in FETCH_SIZE WRITE_SIZE; do
  d=/tmp/pmc_$c; rm -rf $d
  (cd /tmp && timeout 900 rocprofv3 --pmc $c --kernel-trace --kernel-include-regex "corr_mfma" --output-format csv -d $d -o p -- python $root/bench.py $ARGS > /tmp/pmc_$c.log 2>&1)
  echo "$c rc=$?"; tail -2 /tmp/pmc_$c.log | cut -c1-300
  f=$(find $d -name '*counter_collection.csv' | head -1); cp "$f" gpurun_out/r05pmc/$c$SUF.csv
done
python tools/pmc_corr_traffic.py gpurun_out/r05pmc/FETCH_SIZE$SUF.csv gpurun_out/r05pmc/WRITE_SIZE$SUF.csv $K $W $L $I gpurun_out/r05pmc/corr_traffic$SUF.json \
  "rocprofv3 --pmc {FETCH_SIZE|WRITE_SIZE} --kernel-trace --kernel-include-regex corr_mfma -- python bench.py $ARGS" $ROWB "$KNAME"
