#!/usr/bin/env python
"""host-side time per frame of the pipelined tracker, by section (monkeypatched perf_counter wrappers; no GPU syncs
added): tells a host-bound frame (the host never waits in the motion-test read-back) from a GPU-bound one."""
import os, sys, time, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from rampvo_amd import fastba, Ramp_vo as rv
from rampvo_amd.config import make_cfg
from rampvo_amd.synthetic import SyntheticStream, make_network

acc = collections.defaultdict(float)
cnt = collections.defaultdict(int)
stack = []
def wrap(owner, name, tag=None):
    f = getattr(owner, name)
    tag = tag or name
    def g(*a, **k):
        t = time.perf_counter()
        stack.append(0.0)
        try:
            return f(*a, **k)
        finally:
            dt = time.perf_counter() - t
            child = stack.pop()
            acc[tag] += dt - child                 # self time
            acc[tag + " (incl)"] += dt
            cnt[tag] += 1
            if stack:
                stack[-1] += dt
    setattr(owner, name, g)
R = rv.Ramp_vo
for m in ("_track", "_keyframe_finish", "_prefetch_edges", "append_factors", "update", "keyframe", "_graph_plan", "reproject",
          "corr", "_spec_outcome", "_build_next_plan", "_build_plan", "_apply_removal", "settle"):
    wrap(R, m)
wrap(fastba, "BA", "fastba.BA")
wrap(torch.cuda.Event, "synchronize", "Event.synchronize")
import concurrent.futures
wrap(concurrent.futures.Future, "result", "fe job.result")
from rampvo_amd import ops
for m in ("frame_commit", "motionmag"):
    wrap(ops, m, "ops." + m)

net = make_network("SingleScale")
slam = R(make_cfg("default", PATCHES_PER_FRAME=96, MIXED_PRECISION=True), net, {"event_bias": True})
fu_cls = type(net.update.fused(slam.dtype))
for m in ("hidden", "heads_target_weight"):
    wrap(fu_cls, m, "fused." + m)
slam.inputs_ready = True
T, N0 = 400, 200
stream = SyntheticStream(480, 640, T + 1, seed=1234, device="cuda")
frames = [stream.frame(t) for t in range(T)]
torch.cuda.synchronize()
with torch.no_grad():
    for t in range(T):
        if t == N0:
            torch.cuda.synchronize()
            acc.clear(); cnt.clear()
            t_start = time.perf_counter()
        im, ev, K, mask = frames[t]
        slam(t, input_tensor=(ev, im, mask), intrinsics=K)
    slam.settle(); torch.cuda.synchronize()
    wall = time.perf_counter() - t_start
n = T - N0
print("wall %.1f us per frame" % (wall / n * 1e6))
for k in sorted(acc, key=lambda k: -acc[k]):
    print("  %-32s %8.1f us/frame  (%.2f calls)" % (k, acc[k] / n * 1e6, cnt[k.replace(" (incl)", "")] / n))
