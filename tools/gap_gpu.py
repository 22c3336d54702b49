#!/usr/bin/env python
"""GPU-side timeline around the keyframe read-back (HIP events, no profiler): motion test end -> front end done ->
next correlation launch start"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from rampvo_amd import Ramp_vo as rv, ops
from rampvo_amd.config import make_cfg
from rampvo_amd.synthetic import SyntheticStream, make_network
R = rv.Ramp_vo
ev = lambda: torch.cuda.Event(enable_timing=True)
mm_end, corr_start, corr_end, fe_end = [], [], [], []
_mm = ops.motionmag
def mm(*a, **k):
    r = _mm(*a, **k)
    e = ev(); e.record(); mm_end.append(e)
    return r
ops.motionmag = mm
_corr = R._corr_launch
def corr(self, *a, **k):
    s = ev(); s.record(); corr_start.append(s)
    r = _corr(self, *a, **k)
    e = ev(); e.record(); corr_end.append(e)
    return r
R._corr_launch = corr
net = make_network("SingleScale")
_pf = net.patchify.forward
def pf(*a, **k):
    r = _pf(*a, **k)
    e = ev(); e.record(); fe_end.append(e)          # on the stream the front end ran on (the helper thread's current stream)
    return r
net.patchify.forward = pf
slam = R(make_cfg("default", PATCHES_PER_FRAME=96, MIXED_PRECISION=True), net, {"event_bias": True})
slam.inputs_ready = True
T, N0 = 500, 250
stream = SyntheticStream(480, 640, T + 1, seed=1234, device="cuda")
frames = [stream.frame(t) for t in range(T)]
inputs = [(e_, im, mask) for im, e_, K, mask in frames]
torch.cuda.synchronize()
with torch.no_grad():
    for t in range(T):
        if t == N0:
            torch.cuda.synchronize()
            for l in (mm_end, corr_start, corr_end, fe_end):
                l.clear()
        slam(t, input_tensor=inputs[t], intrinsics=frames[t][2])
    slam.settle(); torch.cuda.synchronize()
n = min(len(mm_end), len(corr_start) - 1, len(fe_end) - 1)
# frame i: corr_start[i] .. mm_end[i]; the next frame's front end fe_end[i + 1]; next corr corr_start[i + 1]
chain = np.array([corr_start[i].elapsed_time(mm_end[i]) for i in range(n)]) * 1e3
gap = np.array([mm_end[i].elapsed_time(corr_start[i + 1]) for i in range(n)]) * 1e3
fe_after = np.array([mm_end[i].elapsed_time(fe_end[i + 1]) for i in range(n)]) * 1e3
fe_to_corr = np.array([fe_end[i + 1].elapsed_time(corr_start[i + 1]) for i in range(n)]) * 1e3
pc = lambda a: "p10 %.0f  p50 %.0f  p90 %.0f  mean %.0f" % (np.percentile(a, 10), np.percentile(a, 50), np.percentile(a, 90), a.mean())
print("correlation start -> motion test end (the chain):        ", pc(chain))
print("motion test end -> next correlation start (the gap):     ", pc(gap))
print("motion test end -> next front end done (negative: early):", pc(fe_after))
print("next front end done -> next correlation start:           ", pc(fe_to_corr))
print("frame = chain + gap: %.0f us" % (chain.mean() + gap.mean()))
