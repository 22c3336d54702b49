#!/bin/bash
# tools/sweep_build.sh FILE.hip "-DX=1" "-DX=2" ... -- CMD: rebuild one translation unit per flag set ON THE GPU BOX and run CMD after each
f=$1; shift
flags=()
while [ "$1" != "--" ]; do flags+=("$1"); shift; done
shift
for fl in "${flags[@]}"; do
  touch rampvo_amd/csrc/$f
  make -s -C rampvo_amd/csrc EXTRA="$fl" libramp_hip.so 2>&1 | grep -i "error"
  echo "== $fl"
  "$@"
done
