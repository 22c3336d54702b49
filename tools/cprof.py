#!/usr/bin/env python
"""cProfile of the steady-state tracker loop (host side), top functions by own time per frame"""
import os, sys, cProfile, pstats, io
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from rampvo_amd.config import make_cfg
from rampvo_amd.Ramp_vo import Ramp_vo
from rampvo_amd.synthetic import SyntheticStream, make_network
net = make_network("SingleScale")
slam = Ramp_vo(make_cfg("default", PATCHES_PER_FRAME=96, MIXED_PRECISION=True), net, {"event_bias": True})
slam.inputs_ready = True
T, N0 = 500, 200
stream = SyntheticStream(480, 640, T + 1, seed=1234, device="cuda")
frames = [stream.frame(t) for t in range(T)]
inputs = [(ev, im, mask) for im, ev, K, mask in frames]
torch.cuda.synchronize()
pr = cProfile.Profile()
with torch.no_grad():
    for t in range(T):
        if t == N0:
            pr.enable()
        slam(t, input_tensor=inputs[t], intrinsics=frames[t][2])
    pr.disable()
s = io.StringIO()
ps = pstats.Stats(pr, stream=s).sort_stats("tottime")
ps.print_stats(38)
n = T - N0
for line in s.getvalue().splitlines():
    parts = line.split()
    if len(parts) >= 6 and parts[0].replace("/", "").isdigit():
        try:
            tot, cum = float(parts[1]), float(parts[3])
        except ValueError:
            continue
        print("%8.1f us own %8.1f us cum  x%-6.1f %s" % (tot / n * 1e6, cum / n * 1e6, float(parts[0].split("/")[0]) / n, " ".join(parts[5:])[-95:]))
