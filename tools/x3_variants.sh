#!/bin/bash
# tools/x3_variants.sh "<tag>:<-D flags>" ...: builds csrc/update_x3.hip once per flag set and links each object with the rest of the
# library into build_variants/x3_<tag>/libramp_hip.so (RAMP_HIP_LIB selects one at run time: tools/mb_update_x3.py, bench.py).
# Prints registers / spills of the x3 kernels per variant.  Run in the build container (hipcc cross-compiles).
set -e
cd "$(dirname "$0")/../rampvo_amd/csrc"
make -s -j8 libramp_hip.so >/dev/null 2>&1
OTHERS=$(ls *.o | grep -v update_x3.o)
for spec in "$@"; do
  tag=${spec%%:*}; flags=${spec#*:}
  out=../../build_variants/x3_$tag; mkdir -p $out
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-slp-vectorize -Wall -Wno-unused-function \
      -I../../include $flags -c -o $out/update_x3.o update_x3.hip -save-temps=obj 2>/dev/null
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $out/libramp_hip.so $OTHERS $out/update_x3.o
  echo "== $tag ($flags)"
  grep -E "^\s+\.(name|vgpr_count|vgpr_spill_count|private_segment_fixed_size):" $out/update_x3-hip-amdgcn-amd-amdhsa-gfx950.s \
    | paste - - - - | sed 's/\s\+/ /g' | grep -v segment_softmax
  rm -f $out/*.bc $out/*.hipi $out/*.out $out/*.txt $out/*.hipfb $out/*host* $out/update_x3-hip-amdgcn-amd-amdhsa-gfx950.o
done
