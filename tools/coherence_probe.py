#!/usr/bin/env python
"""producer -> consumer kernel pairs on one stream while a hipGraph of the conv towers replays on another: are writes lost?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from rampvo_amd import conv_hip
from rampvo_amd.synthetic import SyntheticStream, make_network
net = make_network("SingleScale")
enc = net.patchify.encoder
enc.mixed_precision = True
stream = SyntheticStream(480, 640, 4, seed=1, device="cuda")
im, ev, K, mask = stream.frame(1)
with torch.no_grad():
    net.patchify(input_=(ev, im, mask), patches_per_image=96, event_bias=True, reinit_hidden=True)
    s16 = enc._hip_state.ss.view(480, 640, 16)
    fn = lambda: conv_hip.basic_encoder4(enc.imap_encoder, s16, 0.25, half=True)
    fn(); torch.cuda.synchronize()
    G = torch.cuda.CUDAGraph()
    with torch.cuda.graph(G):
        keep = fn()
side = torch.cuda.Stream()
def trial(kind, n=2000, numel=1 << 20):
    x = torch.zeros(numel, device="cuda"); y = torch.zeros(numel, device="cuda")
    bad = torch.zeros((), dtype=torch.int64, device="cuda")
    for i in range(n):
        if kind == "graph":
            with torch.cuda.stream(side):
                G.replay()
        elif kind == "eager":
            with torch.cuda.stream(side), torch.no_grad():
                fn()
        x.fill_(float(i))
        torch.mul(x, 1.0, out=y)
        bad += (y != float(i)).sum()
    torch.cuda.synchronize()
    return int(bad)
for numel in (1 << 14, 1 << 20, 1 << 23):
    print("numel", numel, "| alone:", trial("none", numel=numel), "| eager towers:", trial("eager", 500, numel), "| graph towers:", trial("graph", numel=numel), flush=True)
