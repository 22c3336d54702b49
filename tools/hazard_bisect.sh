#!/bin/bash
# round 6: which half of the victim fails -- the relative pose (packed multiplications / additions only) or the projection
# (divisions)?  SLP builds of tools/mb/pk_hazard.hip with -DVICTIM=1 / 2, next to libramp_hip.so's 1x1 layer (mode 6)
export TMPDIR=/tmp
O=gpurun_out/r06_hazard; mkdir -p $O
B="/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -I include tools/mb/pk_hazard.hip -L rampvo_amd/csrc -lramp_hip"
export LD_LIBRARY_PATH=$PWD/rampvo_amd/csrc:$LD_LIBRARY_PATH
for v in 0 1 2; do $B -DVICTIM=$v -o /tmp/pk_v$v 2>/dev/null || echo "build $v failed"; $B -DVICTIM=$v -fno-slp-vectorize -o /tmp/pk_v${v}_noslp 2>/dev/null; done
for rep in 1 2; do
for v in 0 1 2; do
  echo "== VICTIM=$v, SLP" | tee -a $O/hazard_bisect.txt; timeout 300 /tmp/pk_v$v 6 ${1:-800} 2>&1 | tail -2 | tee -a $O/hazard_bisect.txt
done
done
for v in 1 2; do echo "== VICTIM=$v, -fno-slp-vectorize" | tee -a $O/hazard_bisect.txt; timeout 300 /tmp/pk_v${v}_noslp 6 ${1:-800} 2>&1 | tail -2 | tee -a $O/hazard_bisect.txt; done
