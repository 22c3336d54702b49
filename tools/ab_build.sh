#!/bin/bash
# tools/ab_build.sh "EXTRA flags A" "EXTRA flags B" [rounds] [env...]: rebuilds the whole library with each flag set ON THE GPU BOX
# (two copies of libramp_hip.so), then runs bench.py alternately with each (A B A B ...): one line per run.
A="$1"; B="$2"; R=${3:-2}; shift 3
cd rampvo_amd/csrc
make -s clean; make -s -j 12 EXTRA="$A" libramp_hip.so 2>&1 | grep -i " error" ; cp libramp_hip.so /tmp/libA.so
make -s clean; make -s -j 12 EXTRA="$B" libramp_hip.so 2>&1 | grep -i " error" ; cp libramp_hip.so /tmp/libB.so
cd ../..
for r in $(seq 1 $R); do
  for v in A B; do
    cp /tmp/lib$v.so rampvo_amd/csrc/libramp_hip.so
    bash tools/ab.sh "AB=$v $*"
  done
done
cp /tmp/libA.so rampvo_amd/csrc/libramp_hip.so
