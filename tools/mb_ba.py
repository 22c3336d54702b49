#!/usr/bin/env python
"""microbenchmark of fastba.BA (2 iterations) on a steady-state graph captured from the tracker; per-kernel view via
rocprofv3 --kernel-trace --stats"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from rampvo_amd import fastba
from rampvo_amd.config import make_cfg
from rampvo_amd.Ramp_vo import Ramp_vo
from rampvo_amd.synthetic import SyntheticStream, make_network
net = make_network("SingleScale")
slam = Ramp_vo(make_cfg("default", PATCHES_PER_FRAME=96, MIXED_PRECISION=True), net, {"event_bias": True})
T = 60
stream = SyntheticStream(480, 640, T + 1, seed=100, device="cuda")
cap = {}
inner = fastba.BA
def spy(poses, patches, intrinsics, target, weight, lmbda, ii, jj, kk, t0, t1, *a, **k):
    cap.update(poses=poses.clone(), patches=patches.clone(), intr=intrinsics.clone(), target=target.clone(), weight=weight.clone(),
               lmbda=lmbda, ii=ii.clone(), jj=jj.clone(), kk=kk.clone(), t0=t0, t1=t1, kw=dict(k))
    return inner(poses, patches, intrinsics, target, weight, lmbda, ii, jj, kk, t0, t1, *a, **k)
fastba.BA = spy
with torch.no_grad():
    for t in range(T):
        im, ev, K, mask = stream.frame(t)
        slam(t, input_tensor=(ev, im, mask), intrinsics=K)
fastba.BA = inner
c = cap
def run():
    p, pt = c["poses"].clone(), c["patches"].clone()
    inner(p, pt, c["intr"], c["target"], c["weight"], c["lmbda"], c["ii"], c["jj"], c["kk"], c["t0"], c["t1"], **c["kw"])
    return p, pt
for _ in range(5):
    run()
torch.cuda.synchronize()
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
ps = [(c["poses"].clone(), c["patches"].clone()) for _ in range(50)]
torch.cuda.synchronize()
s.record()
for p, pt in ps:
    inner(p, pt, c["intr"], c["target"], c["weight"], c["lmbda"], c["ii"], c["jj"], c["kk"], c["t0"], c["t1"], **c["kw"])
e.record(); torch.cuda.synchronize()
print("E=%d N=%d RAMP_BA_CHOL=%s: fastba.BA %.1f us per call (2 iterations)" % (c["ii"].shape[0], c["t1"] - c["t0"], os.environ.get("RAMP_BA_CHOL", "0"), s.elapsed_time(e) / 50 * 1e3))
