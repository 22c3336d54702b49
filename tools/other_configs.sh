#!/bin/bash
# bench.py on the other BASELINE configs (parity / cpu legs off): one JSON line each into gpurun_out/$1/
TAG=${1:-r02_cfg}
mkdir -p gpurun_out/$TAG
run() { name=$1; shift; timeout 900 python bench.py --cpu-steps 0 --parity 0 --fp32-leg 0 "$@" 2>/dev/null | grep '^{' > gpurun_out/$TAG/bench_$name.json; python - gpurun_out/$TAG/bench_$name.json $name <<'PY'
import json, sys
d = json.load(open(sys.argv[1])); c = d["config"]
print("%-28s %7.1f kf/s %7.3f ms/step  E=%s  non-pipelined %s" % (sys.argv[2], d["value"], d["ms_per_step"], c.get("edges"), c.get("non_pipelined_kfps")))
PY
}
run multiscale_default --mode MultiScale
run multiscale_precise --mode MultiScale --preset precise --prime 160
run multiscale_720p_m256 --mode MultiScale --height 720 --width 1280 --patches 256 --opt-window 32
run multiscale_720p_m256_fp8 --mode MultiScale --height 720 --width 1280 --patches 256 --opt-window 32 --encoder-fp8 1
run singlescale_fp32 --mixed 0
run singlescale_pipeline0 --pipeline 0
