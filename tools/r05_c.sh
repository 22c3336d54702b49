#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out/r05c
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -o /tmp/chain_tile tools/mb/chain_tile.hip 2>&1 | tail -3
( /tmp/chain_tile 40960 6; /tmp/chain_tile 40960 6 1; /tmp/chain_tile 40960 2 ) > gpurun_out/r05c/chain_tile.txt 2>&1
bash tools/r05_pmc_corr.sh > gpurun_out/r05c/pmc.log 2>&1
tail -40 gpurun_out/r05c/pmc.log
timeout 900 python bench.py > gpurun_out/r05c/bench_default.json 2> gpurun_out/r05c/bench_default.err; echo "bench rc=$?"
