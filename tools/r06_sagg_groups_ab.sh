#!/bin/bash
# round 6: the fp32 pair SoftAgg as one launch without its [f | g] rows, bit-identical (x3_softagg_groups_kernel) -- unit test
# (bit equality with the two launches), fp32 pipeline tests, A/B against a build without it (-DX3_NO_SAGG)
export TMPDIR=/tmp
O=gpurun_out/r06_saggg; mkdir -p $O
timeout 900 python -m pytest tests/test_update_x3_gpu.py -m gpu -q -x > $O/pytest_x3.log 2>&1; echo "x3 rc=$?"; tail -3 $O/pytest_x3.log
timeout 1200 python -m pytest tests/test_pipeline_gpu.py -m gpu -q -x -k "trajectory_fp32 or full_size or device_resident or precise" > $O/pytest_pipe.log 2>&1; echo "pipeline rc=$?"; tail -3 $O/pytest_pipe.log
cd rampvo_amd/csrc
cp libramp_hip.so /tmp/libB.so
rm -f track.o; make -s EXTRA="-DX3_NO_SAGG" libramp_hip.so 2>&1 | grep -i " error"; cp libramp_hip.so /tmp/libA.so
cd ../..
line() { python - "$1" <<'PY'
import json, sys
d = json.load(open("/tmp/l.json")); c = d["config"]
print("%-26s %6.1f kf/s  %.3f ms  corr %.1f us  operator %.1f us (alone %s)  non-pipelined %.1f  converged %.1f" % (
    sys.argv[1], d["value"], d["ms_per_step"], d["roofline"]["mean_launch_us"], d["roofline_update"]["mean_call_us"],
    d["roofline_update"].get("mean_call_us_alone"), c["non_pipelined_kfps"], c["converged_kfps"]))
PY
}
for rep in 1 2 3; do
  for v in A B; do
    cp /tmp/lib$v.so rampvo_amd/csrc/libramp_hip.so
    timeout 300 python bench.py --mixed 0 --cpu-steps 0 --parity 0 2>/dev/null | grep '^{' > /tmp/l.json
    line "$([ $v = A ] && echo 'A: rows + segment softmax' || echo 'B: one launch per group')" | tee -a $O/ab.txt
  done
done
cp /tmp/libB.so rampvo_amd/csrc/libramp_hip.so
