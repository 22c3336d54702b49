#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out/r05h
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -o /tmp/chain_overlap tools/mb/chain_overlap.hip 2>&1 | tail -3
( /tmp/chain_overlap 40960 2; /tmp/chain_overlap 40960 6; /tmp/chain_overlap 40960 4 ) > gpurun_out/r05h/chain_overlap.txt 2>&1; cat gpurun_out/r05h/chain_overlap.txt
timeout 900 python -m pytest -x -q -m gpu tests/test_pipeline_gpu.py -k "full_size_trajectory" -s > gpurun_out/r05h/t_full.log 2>&1; echo "full traj rc=$?"; grep -E "^\.?(True|False) \{" gpurun_out/r05h/t_full.log | cut -c1-700
