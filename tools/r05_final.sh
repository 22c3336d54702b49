#!/bin/bash
# end-of-round deliverables: whole GPU suite, smoke, the default bench line (+ the driver's arguments), kernel table, other configs, PMC passes
TAG=${1:-r05_z}
O=gpurun_out/$TAG
mkdir -p $O
export TMPDIR=/tmp
timeout 2400 python -m pytest -q -m gpu tests/ > $O/gpu_tests.log 2>&1; echo "gpu suite rc=$?"; tail -n 4 $O/gpu_tests.log | cut -c1-200
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?"
timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?"
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_args.json 2> $O/bench_driver_args.err; echo "bench(driver args) rc=$?"
d=/tmp/prof_default
(cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $d -o p -- python $GRAFT_REPO_ROOT/bench.py --cpu-steps 0 --parity 0 --fp32-leg 0 > /tmp/prof_default.out 2>&1)
f=$(find $d -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && python tools/summarize_rocprof.py $f $O/bench_default_kernel_stats.md "bench.py --cpu-steps 0 --parity 0 --fp32-leg 0 (rocprofv3 --kernel-trace --stats)"
grep '^{' /tmp/prof_default.out > $O/bench_under_rocprof.json
python - $O <<'PY'
import json, sys, os
for n in ("bench_default", "bench_driver_args", "bench_under_rocprof"):
    try:
        d = json.loads([l for l in open(os.path.join(sys.argv[1], n + ".json")) if l.startswith("{")][-1])
        print(n, d["value"], d["ms_per_step"], "np", d["config"].get("non_pipelined_kfps"), "conv", d["config"].get("converged_kfps"), "corr", d["roofline"]["mean_launch_us"], d["roofline"]["frac"], "upd", d["roofline_update"]["mean_call_us"], d["roofline_update"].get("mean_call_us_alone"))
    except Exception as e:
        print(n, "FAILED", e)
PY
bash tools/r04_profiles.sh $TAG
