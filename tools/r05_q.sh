#!/bin/bash
mkdir -p gpurun_out/q
for i in 1 2 3 4 5 6 7 8; do
timeout 1200 python -m pytest tests/test_ops_gpu.py tests/test_pipeline_gpu.py -x -q -m gpu -k "ba_ or speculative or device_resident" > gpurun_out/q/t$i.txt 2>&1; echo "run $i rc=$?"; grep -n "AssertionError" gpurun_out/q/t$i.txt | head -3
done
