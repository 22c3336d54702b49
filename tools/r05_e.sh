#!/bin/bash
# same-box A/B of the correlation kernel before / after the corr_edge_rows refactor (+ the tracker-level N2 test)
export TMPDIR=/tmp
mkdir -p gpurun_out/r05e
rm -rf /tmp/prev && mkdir -p /tmp/prev/rampvo_amd && cp -r include /tmp/prev/include && cp -r rampvo_amd/csrc /tmp/prev/rampvo_amd/csrc && cp build_variants/prev/altcorr.hip /tmp/prev/rampvo_amd/csrc/altcorr.hip
(cd /tmp/prev/rampvo_amd/csrc && rm -f altcorr.o libramp_hip.so && make -s libramp_hip.so 2>&1 | grep -i " error")
ls -la /tmp/prev/rampvo_amd/csrc/libramp_hip.so
for r in 1 2 3; do
  echo "prev:"; RAMP_HIP_LIB=/tmp/prev/rampvo_amd/csrc/libramp_hip.so python tools/corr_bench.py 2>&1 | grep "us per call\|Error"
  echo "new:";  python tools/corr_bench.py 2>&1 | grep "us per call\|Error"
done 2>&1 | tee gpurun_out/r05e/corr_ab.txt

B="--cpu-steps 0 --parity 0 --live-steps 0 --np-steps 0"
for r in 1 2; do
for v in prev new; do
  if [ $v = prev ]; then export RAMP_HIP_LIB=/tmp/prev/rampvo_amd/csrc/libramp_hip.so; else unset RAMP_HIP_LIB; fi
  timeout 600 python bench.py $B > /tmp/b_$v.json 2> /tmp/b_$v.err
  python - <<P
import json
d=json.loads([l for l in open('/tmp/b_$v.json') if l.startswith('{')][-1])
print('$v', d['value'], 'corr us', d['roofline']['mean_launch_us'], 'update us', d['roofline_update']['mean_call_us'])
P
done; done 2>&1 | tee -a gpurun_out/r05e/corr_ab.txt
