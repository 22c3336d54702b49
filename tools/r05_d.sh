#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out/r05d
timeout 1200 python -m pytest -x -q -m gpu "tests/test_ops_gpu.py" -k "corr" > gpurun_out/r05d/tests_corr.log 2>&1; echo "corr tests rc=$?"
timeout 900 python -m pytest -x -q -m gpu tests/test_pipeline_gpu.py::test_fused_correlation_linear1_launch_does_not_change_the_tracker > gpurun_out/r05d/tests_trk.log 2>&1; echo "tracker test rc=$?"
tail -5 gpurun_out/r05d/tests_corr.log gpurun_out/r05d/tests_trk.log
B="--cpu-steps 0 --parity 0 --live-steps 0"
for v in "0 4" "1 4" "1 2" "1 8" "0 4"; do set -- $v
  RAMP_CORR_L1=$1 RAMP_CORR_L1_WAVES=$2 timeout 600 python bench.py $B > gpurun_out/r05d/bench_l1_$1_w$2.json 2> gpurun_out/r05d/bench_l1_$1_w$2.err
  python - <<P
import json
d=json.loads([l for l in open('gpurun_out/r05d/bench_l1_$1_w$2.json') if l.startswith('{')][-1])
print('L1=$1 waves=$2', d['value'], 'corr us', d['roofline']['mean_launch_us'], 'update us', d['roofline_update']['mean_call_us'], d['roofline_update']['mean_call_us_alone'], 'np', d['config']['non_pipelined_kfps'])
P
done
