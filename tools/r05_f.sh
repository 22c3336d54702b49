#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out/r05f
timeout 900 python -m pytest -x -q -m gpu tests/test_encoder_gpu.py -k "graph_replay" > gpurun_out/r05f/t1.log 2>&1; echo "graph test rc=$?"; tail -n 3 gpurun_out/r05f/t1.log
timeout 1500 python -m pytest -x -q -m gpu tests/test_pipeline_gpu.py -k "trajectory or device_resident or patchify or intrinsics or front_end" > gpurun_out/r05f/t2.log 2>&1; echo "pipeline tests rc=$?"; tail -n 3 gpurun_out/r05f/t2.log
B="--cpu-steps 0 --parity 0 --live-steps 0 --inst-steps 20"
for r in 1 2; do for f in 0 1; do
  RAMP_SELECT_FORK=$f timeout 600 python bench.py $B > /tmp/b.json 2> /tmp/b.err
  python - <<P
import json
d=json.loads([l for l in open('/tmp/b.json') if l.startswith('{')][-1])
print('fork=$f', d['value'], 'non-pipelined', d['config']['non_pipelined_kfps'], 'front end alone us', d['roofline_encoder'].get('mean_front_end_us'))
P
done; done 2>&1 | tee gpurun_out/r05f/fork_ab.txt
# rocprof table of the fused correlation + Linear1 launch (RAMP_CORR_L1=1)
(cd /tmp && RAMP_CORR_L1=1 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_l1 -o p -- python $GRAFT_REPO_ROOT/bench.py --steps 40 --warmup 5 --cpu-steps 0 --parity 0 --live-steps 0 --np-steps 0 --inst-steps 0 > /tmp/prof_l1.log 2>&1)
f=$(find /tmp/prof_l1 -name '*kernel_stats.csv' | head -1); head -25 "$f" > gpurun_out/r05f/corr_l1_kernel_stats.csv; head -12 gpurun_out/r05f/corr_l1_kernel_stats.csv | cut -c1-160
