#!/usr/bin/env python
"""the live factor count of every steady-state frame (SingleScale 640x480, 96 patches, default.yaml windows): what
DeviceTrack.factor_estimate() and the gru launch's tile choice see"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from rampvo_amd.config import make_cfg
from rampvo_amd.Ramp_vo import Ramp_vo
from rampvo_amd.synthetic import SyntheticStream, make_network
from rampvo_amd import track_dev
slam = Ramp_vo(make_cfg("default", PATCHES_PER_FRAME=96, MIXED_PRECISION=True), make_network("SingleScale"), {"event_bias": True})
slam.inputs_ready = True
N = 400
st = SyntheticStream(480, 640, N + 1, seed=1234, device="cuda")
frames = [st.frame(t) for t in range(N)]
Es = []
for t in range(N):
    im, ev, K, m = frames[t]
    slam(t, input_tensor=(ev, im, m), intrinsics=K)
    if slam._dev is not None and t > 100:
        torch.cuda.synchronize()
        Es.append(int(slam._dev.lazy_state()[track_dev.DYN_E]))
Es = np.array(Es)
print("E: min %d max %d mean %.0f; share <= 40960: %.2f; share <= 40000: %.2f" % (Es.min(), Es.max(), Es.mean(), (Es <= 40960).mean(), (Es <= 40000).mean()))
print(Es[:60].tolist())
