#!/usr/bin/env python
"""bitwise reproducibility of single kernels on STATIC inputs while the imap conv tower replays as a hipGraph on
another stream"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from rampvo_amd import conv_hip, ops, _lib
from rampvo_amd.config import make_cfg
from rampvo_amd.Ramp_vo import Ramp_vo
from rampvo_amd.synthetic import SyntheticStream, make_network
net = make_network("SingleScale")
slam = Ramp_vo(make_cfg("default", PATCHES_PER_FRAME=96, MIXED_PRECISION=True), net, {"event_bias": True})
slam.device_steps = False
T = 50
stream = SyntheticStream(480, 640, T + 1, seed=100, device="cuda")
frames = [stream.frame(t) for t in range(T + 1)]
with torch.no_grad():
    for t in range(T):
        im, ev, K, mask = frames[t]
        slam(t, input_tensor=(ev, im, mask), intrinsics=K)
torch.cuda.synchronize()
enc = net.patchify.encoder
s16 = enc._hip_state.ss.view(480, 640, 16)
with torch.no_grad():
    fn = lambda: conv_hip.basic_encoder4(enc.imap_encoder, s16, 0.25, half=True)
    fn(); torch.cuda.synchronize()
    G = torch.cuda.CUDAGraph()
    with torch.cuda.graph(G):
        keep = fn()
side = torch.cuda.Stream()
E = slam.ii.shape[0]
plan = slam._graph_plan()
coords = slam.reproject()
fu = net.update.fused(torch.float16)
w = fu.weights()
x32 = torch.randn(E, 384, device="cuda")
tests = {
    "transform": lambda: slam.reproject(),
    "corr": lambda: slam.corr(coords, order=plan.g_ij.order),
    "point_cloud": lambda: ops.point_cloud(slam.poses, slam.patches_.view(-1, 3, 3, 3)[:slam.m], slam.intrinsics, slam._ixm[:slam.m]),
    "torch_sin": lambda: torch.sin(x32) * 1.5 + x32,
}
def trial(name, load, n=300):
    f = tests[name]
    with torch.no_grad():
        ref = f().clone()
        torch.cuda.synchronize()
        bad = 0
        for i in range(n):
            if load:
                with torch.cuda.stream(side):
                    G.replay()
            out = f()
            if not torch.equal(torch.nan_to_num(out.float(), nan=-7.0), torch.nan_to_num(ref.float(), nan=-7.0)):
                bad += 1
        torch.cuda.synchronize()
    return bad
for name in tests:
    print(name, "| alone:", trial(name, False, 100), "| with graph load:", trial(name, True), flush=True)
