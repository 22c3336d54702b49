import sys; sys.path.insert(0, '/root/repo')
import torch
from rampvo_amd.config import make_cfg
from rampvo_amd.Ramp_vo import Ramp_vo
from rampvo_amd.synthetic import SyntheticStream, make_network
for seed in (1234, 1235):
    slam = Ramp_vo(make_cfg("default", PATCHES_PER_FRAME=96, MIXED_PRECISION=True), make_network("SingleScale"), {"event_bias": True})
    T = 200
    stream = SyntheticStream(480, 640, T + 1, seed=seed, device="cuda")
    seq = []
    for t in range(T):
        im, ev, K, mask = stream.frame(t)
        n0 = slam.n
        slam(t, input_tensor=(ev, im, mask), intrinsics=K)
        if slam.is_initialized and n0 >= 8: seq.append("R" if slam.n == n0 else "K")
    print(seed, "".join(seq))
