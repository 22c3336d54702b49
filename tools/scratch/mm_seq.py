"""log the keyframe motion-test values and decisions of a run (to study how predictable the decision is)"""
import sys; sys.path.insert(0, '/root/repo')
import torch, numpy as np
from rampvo_amd.config import make_cfg
from rampvo_amd.Ramp_vo import Ramp_vo
from rampvo_amd.synthetic import SyntheticStream, make_network
for seed in (1234, 1235, 1236):
    slam = Ramp_vo(make_cfg("default", PATCHES_PER_FRAME=96, MIXED_PRECISION=True), make_network("SingleScale"), {"event_bias": True})
    T = 300
    stream = SyntheticStream(480, 640, T + 1, seed=seed, device="cuda")
    vals = []
    f0 = slam._keyframe_finish
    def fin():
        r = f0(); h = slam._mm_host.numpy(); vals.append(float((h[0] + h[1]) * 0.5)); return r
    slam._keyframe_finish = fin
    for t in range(T):
        im, ev, K, mask = stream.frame(t)
        slam(t, input_tensor=(ev, im, mask), intrinsics=K)
    print(seed, " ".join("%.2f" % v for v in vals[20:]))
