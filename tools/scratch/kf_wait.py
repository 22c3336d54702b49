"""host time per frame vs. time spent waiting for the keyframe decision; usage: kf_wait.py [pipeline 0|1]"""
import sys, time; sys.path.insert(0, '/root/repo')
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
acc = {"wait": 0.0, "n": 0}
on = [False]
orig = torch.cuda.Event.synchronize
def sync(self):
    t = time.perf_counter(); r = orig(self)
    if on[0]: acc["wait"] += time.perf_counter() - t; acc["n"] += 1
    return r
torch.cuda.Event.synchronize = sync
for t in range(T):
    if t == 80:
        torch.cuda.synchronize(); on[0] = True; t0 = time.perf_counter()
    im, ev, K, mask = frames[t]; slam(t, input_tensor=(ev, im, mask), intrinsics=K)
slam.settle(); torch.cuda.synchronize(); dt = time.perf_counter() - t0
n = T - 80
print("pipeline %d: step %.3f ms; event wait %.3f ms/step (%d waits); host busy %.3f ms/step"
      % (slam.inputs_ready, 1e3 * dt / n, 1e3 * acc["wait"] / n, acc["n"], 1e3 * (dt - acc["wait"]) / n))
