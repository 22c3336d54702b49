#!/bin/bash
# round 6, split-pair fp32 correlation: unit tests, the three-mode A/B of the launch, the fp32 trajectory tests, fp32 bench
export TMPDIR=/tmp
O=gpurun_out/r06x2_a; mkdir -p $O
timeout 600 python -m pytest tests/test_ops_gpu.py -m gpu -q -x -s -k "pyramid_pack or corr" > $O/pytest_ops.log 2>&1; echo "ops rc=$?"; tail -4 $O/pytest_ops.log
grep "level [01]:" $O/pytest_ops.log
timeout 600 python tools/corr_f32_layout.py > $O/corr_f32_layout.txt 2>&1; cat $O/corr_f32_layout.txt | tail -8
timeout 900 python -m pytest tests/test_pipeline_gpu.py -m gpu -q -x -s -k "trajectory_fp32 or (full_size_trajectory and False)" > $O/pytest_traj.log 2>&1; echo "traj rc=$?"; grep -v "^$" $O/pytest_traj.log | tail -12
timeout 600 python bench.py --mixed 0 --cpu-steps 0 --parity 0 > $O/bench_fp32.json 2> $O/bench_fp32.err; echo "bench rc=$?"
python - <<'PY'
import json
for l in open("gpurun_out/r06x2_a/bench_fp32.json"):
    if l.startswith("{"):
        d=json.loads(l); print(d["value"], d["ms_per_step"], {k:d["roofline"].get(k) for k in ("kernel","mean_launch_us","frac")})
PY
