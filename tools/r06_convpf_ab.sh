#!/bin/bash
# round 6: conv_x3_kernel with the next trip's weight fragments requested ahead (CONV_X3_PF) -- encoder tests, A/B on one box
export TMPDIR=/tmp
O=gpurun_out/r06_convpf; mkdir -p $O
timeout 600 python -m pytest tests/test_encoder_gpu.py -m gpu -q -x > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -2 $O/pytest.log
cd rampvo_amd/csrc
cp libramp_hip.so /tmp/libB.so
rm -f conv.o; make -s EXTRA="-DCONV_X3_PF=0" libramp_hip.so 2>&1 | grep -i " error"; cp libramp_hip.so /tmp/libA.so
cd ../..
line() { python - "$1" <<'PY'
import json, sys
d = json.load(open("/tmp/l.json")); c = d["config"]
print("%-14s %6.1f kf/s  %.3f ms  corr %.1f us  operator %.1f us  front end alone %.1f us  non-pipelined %.1f" % (
    sys.argv[1], d["value"], d["ms_per_step"], d["roofline"]["mean_launch_us"], d["roofline_update"]["mean_call_us"],
    d["roofline_encoder"]["mean_front_end_us"], c["non_pipelined_kfps"]))
PY
}
for rep in 1 2 3; do
  for v in A B; do
    cp /tmp/lib$v.so rampvo_amd/csrc/libramp_hip.so
    timeout 300 python bench.py --mixed 0 --cpu-steps 0 --parity 0 2>/dev/null | grep '^{' > /tmp/l.json
    line "$v(PF=$([ $v = A ] && echo 0 || echo 1))" | tee -a $O/ab.txt
  done
done
cp /tmp/libB.so rampvo_amd/csrc/libramp_hip.so
