#!/bin/bash
# tools/mb_variants.sh "python tools/mb_update.py 40000" "" "-DX=1" "-DY=2" ...: rebuilds the library ON THE GPU BOX with each EXTRA
# flag set in turn and runs the given command with it (diagnostic builds of one kernel, microbenchmarks)
CMD="$1"; shift
for F in "$@"; do
  (cd rampvo_amd/csrc && make -s clean && make -s -j 12 EXTRA="$F" libramp_hip.so 2>&1 | grep -i " error")
  echo "== EXTRA='$F'"
  $CMD 2>&1 | tail -${TAIL:-3}
done
(cd rampvo_amd/csrc && make -s clean && make -s -j 12 libramp_hip.so 2>&1 | grep -i " error")
