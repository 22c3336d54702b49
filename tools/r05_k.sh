#!/bin/bash
export TMPDIR=/tmp
B="--cpu-steps 0 --parity 0 --live-steps 0 --np-steps 20 --inst-steps 40"
timeout 900 python -m pytest -x -q -m gpu tests/test_ops_gpu.py -k "ba" 2>&1 | tail -n 2
for r in 1 2; do
for F in "-DBA_PATCH_KEEP_WAVES" ""; do
  (cd rampvo_amd/csrc && rm -f ba.o && make -s EXTRA="$F" libramp_hip.so 2>&1 | grep -i " error")
  timeout 600 python bench.py $B 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); b=d['roofline_ba']; print('[$F]', d['value'], 'BA in frame', b['mean_call_us'], 'alone', b.get('mean_call_us_alone'), 'np', d['config']['non_pipelined_kfps'])"
done; done
(cd rampvo_amd/csrc && rm -f ba.o && make -s libramp_hip.so 2>&1 | grep -i " error")
