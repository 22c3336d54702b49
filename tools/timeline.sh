export TMPDIR=/tmp
cd /tmp && rocprofv3 --kernel-trace --output-format csv -d /tmp/tl -o p -- python $GRAFT_REPO_ROOT/bench.py --steps 150 --warmup 10 --cpu-steps 0 --parity 0 --fp32-leg 0 --np-steps 0 --inst-steps 0 --live-steps 0 2>&1 | grep "^{" > $GRAFT_REPO_ROOT/gpurun_out/tl_bench.json
f=$(find /tmp/tl -name '*kernel_trace.csv' | head -1)
head -1 $f
python $GRAFT_REPO_ROOT/tools/timeline_gaps.py $f 100
