#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out/r05i
timeout 1200 python -m pytest -x -q -m gpu tests/test_pipeline_gpu.py -k "device_resident_steps_and_frame or evaluate_harness or extra_update or pose_prediction" > gpurun_out/r05i/t.log 2>&1; echo "tests rc=$?"; tail -n 12 gpurun_out/r05i/t.log | cut -c1-300
for r in 1 2 3; do
timeout 600 python bench.py --cpu-steps 0 --parity 0 --live-steps 0 > /tmp/b.json 2> /tmp/b.err; tail -2 /tmp/b.err | cut -c1-300
python - <<P
import json
d=json.loads([l for l in open('/tmp/b.json') if l.startswith('{')][-1])
print(d['value'], 'own-stream', d['config'].get('own_stream_kfps'), 'non-pipelined', d['config']['non_pipelined_kfps'])
P
done
