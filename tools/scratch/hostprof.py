"""cProfile of the host side of steady-state frames; usage: hostprof.py [pipeline 0|1]"""
import sys, time, cProfile, pstats; sys.path.insert(0, '/root/repo')
import torch
from rampvo_amd.config import make_cfg
from rampvo_amd.Ramp_vo import Ramp_vo
from rampvo_amd.synthetic import SyntheticStream, make_network
cfg = make_cfg("default", PATCHES_PER_FRAME=96, MIXED_PRECISION=True)
slam = Ramp_vo(cfg, make_network("SingleScale"), {"event_bias": True})
slam.inputs_ready = len(sys.argv) > 1 and sys.argv[1] == "1"
T = 200
stream = SyntheticStream(480, 640, T + 1, seed=1234, device="cuda")
frames = [stream.frame(t) for t in range(T)]
for t in range(80):
    im, ev, K, mask = frames[t]; slam(t, input_tensor=(ev, im, mask), intrinsics=K)
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for t in range(80, T):
    im, ev, K, mask = frames[t]; slam(t, input_tensor=(ev, im, mask), intrinsics=K)
pr.disable()
slam.settle(); torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(45)
st.sort_stats("cumtime").print_stats(60)
