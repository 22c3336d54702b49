#!/usr/bin/env python
"""how long does the main queue wait for the next frame's front end in the pipelined steady state?  Two HIP events around
the main stream's wait for the front end's "done" event (RAMP_FE_WAIT_PROBE=1): the first completes when the previous
frame's plan has, the second when the front end has too.  usage: tools/fe_wait.py [SingleScale|MultiScale] [frames]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rampvo_amd.Ramp_vo as _rv
_rv._FE_WAIT_PROBE = True
import numpy as np, torch
from rampvo_amd.config import make_cfg
from rampvo_amd.Ramp_vo import Ramp_vo
from rampvo_amd.synthetic import SyntheticStream, make_network
mode = sys.argv[1] if len(sys.argv) > 1 else "SingleScale"
T = int(sys.argv[2]) if len(sys.argv) > 2 else 400
torch.manual_seed(1234)
slam = Ramp_vo(make_cfg("default", PATCHES_PER_FRAME=96, MIXED_PRECISION=True), make_network(mode), {"event_bias": True})
slam.inputs_ready = True
slam.fe_wait_pairs = []
st = SyntheticStream(480, 640, T + 1, seed=1234, device="cuda")
frames = [st.frame(t) for t in range(T)]
torch.cuda.synchronize()
with torch.no_grad():
    for t in range(T):
        if t == T - 200:
            torch.cuda.synchronize(); slam.fe_wait_pairs.clear(); t0 = time.perf_counter()
        im, ev, K, mask = frames[t]
        slam(t, input_tensor=(ev, im, mask), intrinsics=K)
    torch.cuda.synchronize()
    per = (time.perf_counter() - t0) / 200 * 1e6
w = np.array([a.elapsed_time(b) * 1e3 for a, b in slam.fe_wait_pairs])
print("%s: %.0f us per frame; main queue waits for the front end %.1f us on average (median %.1f, p90 %.1f; %d frames)" % (
    mode, per, w.mean(), np.median(w), np.percentile(w, 90), len(w)))
