#!/bin/bash
# round 6: the correlation launch scheduled by (target frame, band of window rows) -- equivalence tests, A/B in both precisions
export TMPDIR=/tmp
O=gpurun_out/r06_bands; mkdir -p $O
timeout 1200 python -m pytest tests/test_pipeline_gpu.py -m gpu -q -x -k "device_resident or trajectory_fp32 or full_size_trajectory or hand_back" > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log
f16() { python - "$1" <<'PY'
import json, sys
d = json.load(open("/tmp/l.json")); c = d["config"]; r = d["roofline"]; lc = d.get("roofline_live_compact") or {}; lw = d.get("roofline_live") or {}
print("%-20s %7.1f kf/s  %.3f ms  corr %.1f us  | converged %.1f kf/s corr %.1f us | live %.1f kf/s corr %.1f us | non-pipelined %.1f" % (
    sys.argv[1], d["value"], d["ms_per_step"], r["mean_launch_us"], c["converged_kfps"], lc.get("mean_launch_us", 0), lw.get("kfps", 0), lw.get("mean_launch_us", 0), c["non_pipelined_kfps"]))
PY
}
for rep in 1 2 3; do
  for v in 32 0; do
    RAMP_CORR_BANDS=$v timeout 300 python bench.py --cpu-steps 0 --parity 0 --fp32-leg 0 2>/dev/null | grep '^{' > /tmp/l.json
    f16 "fp16 BANDS=$v" | tee -a $O/ab.txt
  done
done
for rep in 1 2; do
  for v in 32 0; do
    RAMP_CORR_BANDS=$v timeout 300 python bench.py --mixed 0 --cpu-steps 0 --parity 0 2>/dev/null | grep '^{' > /tmp/l.json
    f16 "fp32 BANDS=$v" | tee -a $O/ab.txt
  done
done
for v in 8 16; do
  RAMP_CORR_BANDS=$v timeout 300 python bench.py --cpu-steps 0 --parity 0 --fp32-leg 0 2>/dev/null | grep '^{' > /tmp/l.json
  f16 "fp16 BANDS=$v" | tee -a $O/ab.txt
done
