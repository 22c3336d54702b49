#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out/r05g
timeout 900 python -m pytest -x -q -m gpu tests/test_pipeline_gpu.py -k "full_size_trajectory" -s > gpurun_out/r05g/t_full.log 2>&1; echo "full traj rc=$?"; grep -E "^(True|False) \{|passed|failed|Error" gpurun_out/r05g/t_full.log | cut -c1-600
timeout 2400 python -m pytest -x -q -m gpu tests/ > gpurun_out/r05g/gpu_tests.log 2>&1; echo "gpu suite rc=$?"; tail -n 6 gpurun_out/r05g/gpu_tests.log | cut -c1-300
