#!/bin/bash
# end-of-round deliverables in one gpurun call: the other BASELINE configs (JSON lines + kernel tables) and the PMC passes
# (MFMA utilisation per config, correlation traffic).  usage: tools/r04_profiles.sh TAG
TAG=${1:-r04_z}
O=gpurun_out/$TAG
mkdir -p $O
export TMPDIR=/tmp
line() { name=$1; shift; timeout 1200 python bench.py --cpu-steps 0 --parity 0 --fp32-leg 0 "$@" 2>$O/${name}.err | grep '^{' > $O/bench_${name}.json; python - $O/bench_${name}.json $name <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1])); c = d["config"]
    print("%-30s %7.1f kf/s %7.3f ms/step  E=%s  np %s  | %s" % (sys.argv[2], d["value"], d["ms_per_step"], c.get("edges"), c.get("non_pipelined_kfps"), c["workload"][:70]))
except Exception as e:
    print(sys.argv[2], "FAILED", e)
PY
}
prof() { name=$1; shift; d=/tmp/prof_$name; (cd /tmp && timeout 1200 rocprofv3 --kernel-trace --stats --output-format csv -d $d -o p -- python $GRAFT_REPO_ROOT/bench.py --cpu-steps 0 --parity 0 --fp32-leg 0 --live-steps 0 --np-steps 0 --inst-steps 0 --steps 60 "$@" > /tmp/prof_$name.out 2>&1); f=$(find $d -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && python tools/summarize_rocprof.py $f $O/kernel_stats_${name}.md "bench.py --steps 60 $* (rocprofv3 --kernel-trace --stats)"; }
line multiscale_default --mode MultiScale
line config2_multiscale_precise --config 2
line config2_full_window --config 2 --keyframe-thresh 0 --prime 90 --clock-warm-max 60 --inst-steps 40 --np-steps 20
line config4_as_written --config 4
line config4_f16_mfma --config 4 --encoder-fp8 0
line config4_trackers_own_window --config 4 --keyframe-thresh 15 --prime 70
line singlescale_fp32 --mixed 0
line singlescale_pipeline0 --pipeline 0
prof multiscale_default --mode MultiScale
prof config2_multiscale_precise --config 2
prof config4_as_written --config 4
# PMC: MFMA busy + GUI active in one pass per config; correlation traffic (FETCH / WRITE in their own passes)
STEPS=6 PRIME=64 bash tools/pmc_passes.sh $O/pmc_default "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" "FETCH_SIZE" "WRITE_SIZE" > /dev/null 2>&1
STEPS=6 PRIME=64 BENCH_ARGS="--mode MultiScale" bash tools/pmc_passes.sh $O/pmc_multiscale_default "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" > /dev/null 2>&1
STEPS=6 PRIME=120 BENCH_ARGS="--config 2" bash tools/pmc_passes.sh $O/pmc_config2 "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" > /dev/null 2>&1
STEPS=6 PRIME=60 BENCH_ARGS="--config 4" bash tools/pmc_passes.sh $O/pmc_config4 "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" > /dev/null 2>&1
for c in default multiscale_default config2 config4; do
  [ -f $O/pmc_$c/pass_1.txt ] && python tools/mfma_util.py $O/pmc_$c/pass_1.txt "MFMA pipe utilisation per kernel, $TAG, $c" > $O/mfma_utilisation_$c.md
done
ls $O
