#!/bin/bash
# round 5, first GPU call: the tile-shape microbenchmark, the new parity tests, a baseline bench line
export TMPDIR=/tmp
mkdir -p gpurun_out/r05a
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -o /tmp/chain_tile tools/mb/chain_tile.hip 2>&1 | tail -3
for args in "40000 6" "40960 6" "40000 2"; do /tmp/chain_tile $args; done > gpurun_out/r05a/chain_tile.txt 2>&1
timeout 1500 python -m pytest -x -q -m gpu tests/test_integration_shims_gpu.py "tests/test_pipeline_gpu.py::test_full_size_update_step_in_the_wide_regime" tests/test_pipeline_gpu.py::test_bench_under_torchrun_initialises_rccl -s > gpurun_out/r05a/new_tests.log 2>&1
echo "new tests rc=$?" >> gpurun_out/r05a/new_tests.log
timeout 900 python bench.py > gpurun_out/r05a/bench_default.json 2> gpurun_out/r05a/bench_default.err
echo "bench rc=$?"
tail -c 1500 gpurun_out/r05a/new_tests.log
cat gpurun_out/r05a/chain_tile.txt
