#!/bin/bash
# end-of-round deliverables (round 6): whole GPU suite, smoke, the default bench line (+ the driver's arguments), kernel tables
# (fp16 default and --mixed 0), the other configs, PMC passes (MFMA utilisation, correlation traffic, the x3 chains)
TAG=${1:-r06_z}
O=gpurun_out/$TAG
mkdir -p $O
export TMPDIR=/tmp
(time timeout 2400 python -m pytest -q -m gpu tests/) > $O/gpu_tests.log 2>&1; echo "gpu suite rc=$?"; tail -n 8 $O/gpu_tests.log | cut -c1-200
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?"
timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?"
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_args.json 2> $O/bench_driver_args.err; echo "bench(driver args) rc=$?"
for v in default fp32; do
  d=/tmp/prof_$v; extra=""; [ $v = fp32 ] && extra="--mixed 0"
  (cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $d -o p -- python $GRAFT_REPO_ROOT/bench.py --cpu-steps 0 --parity 0 --fp32-leg 0 $extra > /tmp/prof_$v.out 2>&1)
  f=$(find $d -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && python tools/summarize_rocprof.py $f $O/bench_${v}_kernel_stats.md "bench.py --cpu-steps 0 --parity 0 --fp32-leg 0 $extra (rocprofv3 --kernel-trace --stats)"
  grep '^{' /tmp/prof_$v.out > $O/bench_${v}_under_rocprof.json
done
python - $O <<'PY'
import json, sys, os
for n in ("bench_default", "bench_driver_args", "bench_default_under_rocprof", "bench_fp32_under_rocprof"):
    try:
        d = json.loads([l for l in open(os.path.join(sys.argv[1], n + ".json")) if l.startswith("{")][-1])
        c = d["config"]
        print(n, d["value"], d["ms_per_step"], "np", c.get("non_pipelined_kfps"), "conv", c.get("converged_kfps"), "fp32", c.get("fp32_kfps"), "corr", d["roofline"]["mean_launch_us"], d["roofline"]["frac"], "upd", d["roofline_update"]["mean_call_us"], d["roofline_update"].get("mean_call_us_alone"))
    except Exception as e:
        print(n, "FAILED", e)
PY
bash tools/converged_repeats.sh 3 > $O/converged_repeats.txt 2>&1; cat $O/converged_repeats.txt
timeout 900 python bench.py --preset precise --patches 300 --cpu-steps 0 --parity 0 --steps 60 --prime 90 --clock-warm-max 60 --inst-steps 30 --np-steps 10 --live-steps 0 --fp32-leg 0 2>$O/bench_precise300.err | grep "^{" > $O/bench_precise_300_patches.json
bash tools/pmc_x3.sh $O/pmc_x3 > /dev/null 2>&1
bash tools/r05_pmc_corr.sh > $O/pmc_corr.log 2>&1; cp gpurun_out/r05pmc/corr_traffic.json $O/corr_traffic.json 2>/dev/null
MIXED=0 bash tools/r05_pmc_corr.sh > $O/pmc_corr_fp32.log 2>&1; cp gpurun_out/r05pmc/corr_traffic_fp32.json $O/corr_traffic_fp32.json 2>/dev/null
bash tools/r04_profiles.sh $TAG
