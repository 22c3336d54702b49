#!/bin/bash
# one gpurun call: GPU tests, a default bench run, a kernel-trace profile of a short bench run
# usage (on the GPU box, from the repo root): tools/gpu_round.sh TAG [pytest -k expr]
TAG=${1:-r02_a}
KEXPR=${2:-}
export TMPDIR=/tmp
mkdir -p gpurun_out/$TAG
if [ -n "$KEXPR" ]; then
  timeout 1500 python -m pytest tests -m gpu -q -x -k "$KEXPR" -s > gpurun_out/$TAG/pytest.log 2>&1
else
  timeout 1500 python -m pytest tests -m gpu -q -s > gpurun_out/$TAG/pytest.log 2>&1
fi
echo "pytest rc=$?" >> gpurun_out/$TAG/pytest.log
tail -5 gpurun_out/$TAG/pytest.log
timeout 600 python bench.py > gpurun_out/$TAG/bench.json 2> gpurun_out/$TAG/bench.err
echo "bench rc=$?"; tail -c 3000 gpurun_out/$TAG/bench.json
timeout 600 python bench.py --steps 20 --warmup 5 > gpurun_out/$TAG/bench_driver_args.json 2>> gpurun_out/$TAG/bench.err
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$TAG -o p -- python $GRAFT_REPO_ROOT/bench.py --steps 100 --warmup 10 --cpu-steps 0 --parity 0 --fp32-leg 0 --np-steps 0 > /tmp/prof_$TAG.log 2>&1)
f=$(find /tmp/prof_$TAG -name '*kernel_stats.csv' | head -1)
[ -n "$f" ] && python tools/summarize_rocprof.py $f gpurun_out/$TAG/kernel_stats.md "$TAG: bench.py --steps 100 --warmup 10 (default config)"
grep -h "^{" /tmp/prof_$TAG.log > gpurun_out/$TAG/bench_under_rocprof.json
