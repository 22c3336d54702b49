#!/bin/bash
# round 6: correlation waves fed by schedule-ordered records (ramp_track.corr_desc) -- equivalence tests, A/B, both precisions
export TMPDIR=/tmp
O=gpurun_out/r06_desc; mkdir -p $O
timeout 900 python -m pytest tests/test_pipeline_gpu.py -m gpu -q -x -k "device_resident or trajectory_fp32 or full_size_update_step_against" > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log
for rep in 1 2 3; do
  bash tools/ab.sh "RAMP_CORR_DESC=1" "RAMP_CORR_DESC=0" | tee -a $O/ab_f16.txt
done
for rep in 1 2; do
for v in 1 0; do
  RAMP_CORR_DESC=$v timeout 300 python bench.py --mixed 0 --cpu-steps 0 --parity 0 2>/dev/null | grep '^{' > /tmp/l.json
  python - $v <<'PY' | tee -a $O/ab_f32.txt
import json, sys
d = json.load(open("/tmp/l.json"))
print("fp32 RAMP_CORR_DESC=%s  %6.1f kf/s  %.3f ms  corr %.1f us  converged %.1f" % (sys.argv[1], d["value"], d["ms_per_step"], d["roofline"]["mean_launch_us"], d["config"]["converged_kfps"]))
PY
done
done
python - <<'PY' | tee $O/corr_us.txt
import json
PY
