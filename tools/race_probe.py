#!/usr/bin/env python
"""long pipelined runs WITHOUT per-frame synchronisation; reports whether the tracker state is still finite at the end.
usage: tools/race_probe.py [--src DIR] [--runs N] [--frames T]   (--src: a tree holding another rampvo_amd/)"""
import argparse, os, sys
ap = argparse.ArgumentParser()
ap.add_argument("--src", default=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ap.add_argument("--runs", type=int, default=4)
ap.add_argument("--frames", type=int, default=1200)
ap.add_argument("--pipelined", type=int, default=1)
ap.add_argument("--w-bias", type=float, default=None)
ap.add_argument("--seed0", type=int, default=100)
ap.add_argument("--seeded-depth", type=int, default=0)
a = ap.parse_args()
sys.path.insert(0, a.src)
import torch
from rampvo_amd.config import make_cfg
from rampvo_amd.Ramp_vo import Ramp_vo
from rampvo_amd.synthetic import SyntheticStream, make_network
import rampvo_amd
print("package:", os.path.dirname(rampvo_amd.__file__))
for r in range(a.runs):
    kw = {} if a.w_bias is None else {"w_bias": a.w_bias}
    net = make_network("SingleScale", **kw)
    slam = Ramp_vo(make_cfg("default", PATCHES_PER_FRAME=96, MIXED_PRECISION=True), net, {"event_bias": True})
    slam.inputs_ready = bool(a.pipelined)
    fno = [0]
    if a.seeded_depth:
        def draw(patches):
            g = torch.Generator().manual_seed(4321 + fno[0])
            return torch.rand(1, patches.shape[1], 1, 1, generator=g).to(patches.device)
        slam._initial_depth = draw
    stream = SyntheticStream(480, 640, a.frames + 1, seed=a.seed0 + r, device="cuda")
    frames = [stream.frame(t) for t in range(a.frames)]
    torch.cuda.synchronize()
    first_bad = None
    with torch.no_grad():
        for t in range(a.frames):
            im, ev, K, mask = frames[t]
            fno[0] = t
            slam(t, input_tensor=(ev, im, mask), intrinsics=K)
            if t % 100 == 99 and first_bad is None and len(slam._ii) >= 45312:
                first_bad = t
    slam.settle()
    torch.cuda.synchronize()
    n = slam.n
    chk = float(slam.poses_[:n].double().nan_to_num(0).abs().sum())
    print("seed %d: n=%d E=%d chk=%.9g finite=%s first E==bound at t=%s" % (a.seed0 + r, n, len(slam._ii), chk, bool(torch.isfinite(slam.poses_[:n]).all()), first_bad), flush=True)
    del slam, net, frames, stream
