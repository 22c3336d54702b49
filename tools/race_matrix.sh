#!/bin/bash
# pipelined-vs-sequential divergence detector: same seeds under different overlap toggles
R=${1:-5}
echo "== sequential"; python tools/race_probe.py --runs $R --pipelined 0 2>&1 | grep "^seed"
echo "== pipelined"; python tools/race_probe.py --runs $R 2>&1 | grep "^seed"
echo "== pipelined, no helper thread"; RAMP_FE_THREAD=0 python tools/race_probe.py --runs $R 2>&1 | grep "^seed"
echo "== pipelined, no hipGraph"; RAMP_NO_GRAPH=1 python tools/race_probe.py --runs $R 2>&1 | grep "^seed"
echo "== pipelined, FE at ba"; RAMP_FE_AT=ba python tools/race_probe.py --runs $R 2>&1 | grep "^seed"
