#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out/r05b
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -o /tmp/chain_tile tools/mb/chain_tile.hip 2>&1 | tail -3
( /tmp/chain_tile 40960 6 | tail -12; /tmp/chain_tile 40960 6 1 | tail -12 ) > gpurun_out/r05b/chain_tile2.txt 2>&1
cut -c1-150 gpurun_out/r05b/chain_tile2.txt
