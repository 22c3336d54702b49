#!/bin/bash
# round 6: conv_x3_kernel's LDS padding (CONV_X3_PAD: 16 bytes per pixel and plane; 0: 112 -> 75 KB for the 7x7 layer, so that an
# operator workgroup fits beside it) -- fp32 bench alternately, same box
export TMPDIR=/tmp
O=gpurun_out/r06_convpad; mkdir -p $O
cd rampvo_amd/csrc
cp libramp_hip.so /tmp/libA.so
rm -f conv.o; make -s EXTRA="-DCONV_X3_PAD=0" libramp_hip.so 2>&1 | grep -i " error"; cp libramp_hip.so /tmp/libB.so
rm -f conv.o; make -s EXTRA="-DCONV_X3_PAD=8" libramp_hip.so 2>&1 | grep -i " error"; cp libramp_hip.so /tmp/libC.so
cd ../..
line() { python - "$1" <<'PY'
import json, sys
d = json.load(open("/tmp/l.json")); c = d["config"]
print("%-10s %6.1f kf/s  %.3f ms  corr %.1f us  operator %.1f us (alone %s)  front end alone %.1f us  non-pipelined %.1f" % (
    sys.argv[1], d["value"], d["ms_per_step"], d["roofline"]["mean_launch_us"], d["roofline_update"]["mean_call_us"],
    d["roofline_update"].get("mean_call_us_alone"), d["roofline_encoder"]["mean_front_end_us"], c["non_pipelined_kfps"]))
PY
}
for rep in 1 2 3; do
  for v in A B C; do
    cp /tmp/lib$v.so rampvo_amd/csrc/libramp_hip.so
    timeout 300 python bench.py --mixed 0 --cpu-steps 0 --parity 0 2>/dev/null | grep '^{' > /tmp/l.json
    line "$([ $v = A ] && echo 'PAD=16' || ([ $v = B ] && echo 'PAD=0' || echo 'PAD=8'))" | tee -a $O/ab.txt
  done
done
cp /tmp/libA.so rampvo_amd/csrc/libramp_hip.so
