#!/bin/bash
mkdir -p gpurun_out/o
timeout 1200 python -m pytest tests/test_ops_gpu.py -x -q -m gpu -k "ba_" > gpurun_out/o/ba_tests.txt 2>&1; echo "ba tests rc=$?"; tail -5 gpurun_out/o/ba_tests.txt
bash tools/tail_kernels.sh "" new 2>&1 | grep "ba_\|period"
