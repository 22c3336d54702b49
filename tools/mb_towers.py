#!/usr/bin/env python
"""the two BasicEncoder4 towers alone (10 launches per frame behind the LSTM step): HIP events around 50 eager passes and
around 50 replays of a hipGraph of one pass.  With tools/mb_variants.sh and -DHZ_SKIP_MMA / -DHZ_SKIP_EPI: what the MFMA
phase and the epilogue of the LDS-tiled conv kernel cost."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from rampvo_amd import conv_hip
from rampvo_amd.synthetic import make_network
mode = sys.argv[1] if len(sys.argv) > 1 else "SingleScale"
H, W = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (480, 640)
net = make_network(mode).cuda()
enc = net.patchify.encoder
g = torch.Generator().manual_seed(3)
rnd = lambda *s: (torch.randn(*s, generator=g) * 0.5).cuda()
encs = [enc.fmap_encoder, enc.imap_encoder]
if mode == "SingleScale":
    x = rnd(H, W, 16)
    fn = lambda: conv_hip.basic_encoder4_towers(encs, x, 0.25, half=True)
else:
    x, x2, x4 = rnd(H, W, 16), rnd(H // 2, W // 2, 32).half(), rnd(H // 4, W // 4, 64).half()
    fn = lambda: conv_hip.multiscale_encoder4_towers(encs, x, x2, x4, 0.25, half=True)
with torch.no_grad():
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(50):
        fn()
    b.record(); torch.cuda.synchronize()
    eager = a.elapsed_time(b) / 50 * 1e3
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        fn(); torch.cuda.synchronize()
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr, stream=s):
            out = fn()
        for _ in range(5):
            gr.replay()
        torch.cuda.synchronize()
        a.record(s)
        for _ in range(50):
            gr.replay()
        b.record(s); torch.cuda.synchronize()
    print("%s towers %dx%d: eager %.1f us, graph replay %.1f us per pass" % (mode, W, H, eager, a.elapsed_time(b) / 50 * 1e3), flush=True)
