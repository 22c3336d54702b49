#!/usr/bin/env python
"""Which kernel on the OTHER stream makes an SLP-vectorised (packed-fp32) kernel return wrong values?
Run with a library whose lie.hip / ba.hip were built WITHOUT -fno-slp-vectorize (RAMP_HIP_LIB=...): the kernel under
test is pops.transform on static inputs, compared bit for bit with its first run, while ONE kind of front-end kernel
replays from a hipGraph on a side stream (8 launches per replay)."""
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
T = 40
stream = SyntheticStream(480, 640, T + 1, seed=100, device="cuda")
with torch.no_grad():
    for t in range(T):
        im, ev, K, mask = stream.frame(t)
        slam(t, input_tensor=(ev, im, mask), intrinsics=K)
torch.cuda.synchronize()
enc = net.patchify.encoder
im_enc = enc.imap_encoder
s16 = enc._hip_state.ss.view(480, 640, 16)
x32h = torch.randn(240, 320, 32, device="cuda").half()
x64h = torch.randn(120, 160, 64, device="cuda").half()
xf = torch.randn(1 << 22, device="cuda")
a16 = torch.randn(4096, 4096, device="cuda").half()
loads = {
    "whole imap tower": lambda: conv_hip.basic_encoder4(im_enc, s16, 0.25, half=True),
    "conv 7x7 s2 (fp32 in, f16 MFMA)": lambda: conv_hip.conv2d(s16, im_enc.conv1, relu=True, half=True),
    "conv 3x3 32ch (f16 MFMA, LDS tiles)": lambda: conv_hip.conv2d(x32h, im_enc.layer1[0].conv1, relu=True, half=True),
    "conv 3x3 64ch": lambda: conv_hip.conv2d(x64h, im_enc.layer2[1].conv1, relu=True, half=True),
    "conv 1x1 64->384": lambda: conv_hip.conv2d(x64h, im_enc.conv2, half=True),
    "torch elementwise (VALU only)": lambda: torch.sin(xf) * 1.5 + xf,
    "hipBLASLt f16 GEMM 4096^3 (MFMA)": lambda: a16 @ a16,
}
side = torch.cuda.Stream()
graphs = {}
with torch.no_grad():
    for name, fn in loads.items():
        fn(); torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            keep = [fn() for _ in range(4 if "tower" in name or "GEMM" in name else 8)]
        graphs[name] = (g, keep)
coords_ref = slam.reproject().clone()
torch.cuda.synchronize()
def trial(g, n=300):
    bad = 0
    with torch.no_grad():
        for i in range(n):
            if g is not None:
                with torch.cuda.stream(side):
                    g.replay()
            out = slam.reproject()
            bad += not torch.equal(out, coords_ref)
        torch.cuda.synchronize()
    return bad
print("lib:", os.environ.get("RAMP_HIP_LIB", "default"))
print("%-40s wrong launches of transform / 300" % "load on the side stream")
print("%-40s %d" % ("(none)", trial(None)))
for name, (g, _) in graphs.items():
    print("%-40s %d" % (name, trial(g)), flush=True)
