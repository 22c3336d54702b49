#!/bin/bash
# round 6: is the packed-fp32 failure a matter of pipeline distance?  The victim (tools/mb/pk_hazard.hip's
# transform kernel, SLP-vectorised: v_pk_*_f32) rebuilt with s_nop padding in front of every instruction and with every wait
# forced to zero, next to libramp_hip.so's 1x1 conv layer replayed from a hipGraph (mode 6)
export TMPDIR=/tmp
O=gpurun_out/r06_hazard; mkdir -p $O
B="/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -I include tools/mb/pk_hazard.hip -L rampvo_amd/csrc -lramp_hip"
export LD_LIBRARY_PATH=$PWD/rampvo_amd/csrc:$LD_LIBRARY_PATH
build() { name=$1; shift; $B "$@" -o /tmp/pk_$name 2>/dev/null || echo "build $name failed"; }
build slp
build noslp -fno-slp-vectorize
build nop1 -mllvm -amdgpu-snop-padding=1
build nop3 -mllvm -amdgpu-snop-padding=3
build nop7 -mllvm -amdgpu-snop-padding=7
build nop15 -mllvm -amdgpu-snop-padding=15
build wait0 -mllvm -amdgpu-waitcnt-forcezero
R=${1:-600}
for rep in 1 2; do
for v in slp noslp nop1 nop3 nop7 nop15 wait0; do
  [ -x /tmp/pk_$v ] || continue
  echo "== $v (mode 6, $R rounds x 4 launches)" | tee -a $O/hazard_nops.txt
  timeout 300 /tmp/pk_$v 6 $R 2>&1 | tail -3 | tee -a $O/hazard_nops.txt
done
done
