#!/bin/bash
# round 6: (1) where the fp32 front end may start, with the split-pair correlation in the step; (2) register budget / loads in
# flight of corr_mfma_kernel<CorrX2> (altcorr.o rebuilt on the box per variant, the launch alone on the tracker's own graph)
export TMPDIR=/tmp
O=gpurun_out/r06x2_b; mkdir -p $O
line() { python - "$1" "$2" <<'PY'
import json, sys
for l in open(sys.argv[2]):
    if l.startswith("{"):
        d = json.loads(l); c = d["config"]
        print("%-28s %6.1f kf/s  %.3f ms  corr %.1f us  operator %.1f us  front end %.1f us  non-pipelined %.1f  converged %.1f" % (
            sys.argv[1], d["value"], d["ms_per_step"], d["roofline"]["mean_launch_us"], d["roofline_update"]["mean_call_us"],
            d["roofline_encoder"]["mean_front_end_us"], c["non_pipelined_kfps"], c["converged_kfps"]))
PY
}
for rep in 1 2; do
for g in 5 4 3 2; do
  RAMP_GATE_AT_F32=$g timeout 300 python bench.py --mixed 0 --cpu-steps 0 --parity 0 > $O/gate$g.json 2> $O/gate$g.err
  line "RAMP_GATE_AT_F32=$g" $O/gate$g.json | tee -a $O/gate_ab.txt
done
done
cd rampvo_amd/csrc
cp libramp_hip.so /tmp/lib_default.so
for v in "2 3" "2 4" "3 3" "4 3" "1 4"; do
  set -- $v
  rm -f altcorr.o
  make -s EXTRA="-DCORR_X2_PGB=$1 -DCORR_X2_WAVES=$2" libramp_hip.so 2>&1 | grep -i " error"
  echo "== CORR_X2_PGB=$1 CORR_X2_WAVES=$2" | tee -a ../../$O/x2_variants.txt
  (cd ../.. && CORR_F32_MODES=22 timeout 300 python tools/corr_f32_layout.py 2>&1 | grep "us per launch" | tee -a $O/x2_variants.txt)
done
cp /tmp/lib_default.so libramp_hip.so
