"""GPU-side decomposition of the read-back gap with timing events on the main stream (pipelined mode)"""
import sys, time; sys.path.insert(0, '/root/repo')
import torch, numpy as np
from rampvo_amd import ops
from rampvo_amd.config import make_cfg
from rampvo_amd.Ramp_vo import Ramp_vo
from rampvo_amd.synthetic import SyntheticStream, make_network
slam = Ramp_vo(make_cfg("default", PATCHES_PER_FRAME=96, MIXED_PRECISION=True), make_network("SingleScale"), {"event_bias": True})
slam.inputs_ready = True
T = 240
stream = SyntheticStream(480, 640, T + 1, seed=1234, device="cuda")
frames = [stream.frame(t) for t in range(T)]
on = [False]; marks = {k: [] for k in ("mm", "finish_in", "finish_out", "commit_in", "commit_out", "fe_wait_out", "corr_in")}
def ev(name):
    if on[0]:
        e = torch.cuda.Event(enable_timing=True); e.record(); marks[name].append(e)
def wrap_after(obj, name, mark):
    f = getattr(obj, name)
    def g(*a, **k):
        r = f(*a, **k); ev(mark); return r
    setattr(obj, name, g)
def wrap_before(obj, name, mark):
    f = getattr(obj, name)
    def g(*a, **k):
        ev(mark); return f(*a, **k)
    setattr(obj, name, g)
wrap_after(ops, "motionmag", "mm")
f0 = slam._keyframe_finish
def fin():
    r = f0(); ev("finish_out"); return r
slam._keyframe_finish = fin
fc = ops.frame_commit
def fcw(*a, **k):
    ev("commit_in"); r = fc(*a, **k); ev("commit_out"); return r
ops.frame_commit = fcw
wrap_before(slam, "_corr_launch", "corr_in")
for t in range(T):
    if t == 100:
        torch.cuda.synchronize(); on[0] = True
    im, evs, K, mask = frames[t]; slam(t, input_tensor=(evs, im, mask), intrinsics=K)
slam.settle(); torch.cuda.synchronize()
n = min(len(v) for k, v in marks.items() if v) - 1
def seg(a, b, shift=0):
    d = np.array([marks[a][i].elapsed_time(marks[b][i + shift]) for i in range(n)]) * 1e3
    return "%-26s mean %6.1f  median %6.1f  p90 %6.1f" % (a + " -> " + b, d.mean(), np.median(d), np.percentile(d, 90))
print(seg("mm", "finish_out", 1)); print(seg("finish_out", "commit_in")); print(seg("commit_in", "commit_out"))
print(seg("commit_out", "corr_in")); print(seg("mm", "corr_in", 1))
