#!/usr/bin/env python
"""host time from the motion-test read-back to the next correlation launch, split by the calls in between"""
import os, sys, time, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from rampvo_amd import Ramp_vo as rv, ops, fastba
from rampvo_amd.config import make_cfg
from rampvo_amd.synthetic import SyntheticStream, make_network
R = rv.Ramp_vo
marks = []
state = {"on": False, "t0": None}
def wrap(owner, name, tag):
    f = getattr(owner, name)
    def g(*a, **k):
        t = time.perf_counter()
        try:
            return f(*a, **k)
        finally:
            if state["on"] and state["t0"] is not None:
                marks.append((tag, t - state["t0"], time.perf_counter() - state["t0"]))
    setattr(owner, name, g)
_sync = torch.cuda.Event.synchronize
def sync(self):
    r = _sync(self)
    state["t0"] = time.perf_counter()
    return r
torch.cuda.Event.synchronize = sync
_corr = R._corr_launch
def corr(self, *a, **k):
    r = _corr(self, *a, **k)
    if state["on"] and state["t0"] is not None:
        marks.append(("corr launched", None, time.perf_counter() - state["t0"]))
        state["t0"] = None
    return r
R._corr_launch = corr
for owner, name in ((R, "_apply_removal"), (R, "_prefetch_edges"), (R, "append_factors"), (R, "reproject"), (R, "_graph_plan"), (R, "_wait_upload_stream"), (ops, "frame_commit")):
    wrap(owner, name, name)
import concurrent.futures
wrap(concurrent.futures.Future, "result", "fe job.result")
net = make_network("SingleScale")
slam = R(make_cfg("default", PATCHES_PER_FRAME=96, MIXED_PRECISION=True), net, {"event_bias": True})
slam.inputs_ready = True
T, N0 = 400, 200
stream = SyntheticStream(480, 640, T + 1, seed=1234, device="cuda")
frames = [stream.frame(t) for t in range(T)]
torch.cuda.synchronize()
with torch.no_grad():
    for t in range(T):
        state["on"] = t >= N0
        im, ev, K, mask = frames[t]
        slam(t, input_tensor=(ev, im, mask), intrinsics=K)
    slam.settle(); torch.cuda.synchronize()
n = T - N0
agg = collections.OrderedDict()
for tag, t_in, t_out in marks:
    a = agg.setdefault(tag, [0, 0.0, 0.0])
    a[0] += 1; a[1] += (t_in or 0.0); a[2] += t_out
print("host time after the read-back returns (us, mean over %d frames): call | entered at | returned at" % n)
for tag, (c, a, b) in agg.items():
    print("  %-22s x%.2f  %7.1f  %7.1f" % (tag, c / n, a / c * 1e6, b / c * 1e6))
