#!/usr/bin/env python
"""the MultiScale front end's per-scale LSTM / super-state launches alone (HIP events), 640x480 and 1280x720"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from rampvo_amd import conv_hip
from rampvo_amd.synthetic import make_network
enc = make_network("MultiScale").patchify.encoder
for H, W in ((480, 640), (720, 1280)):
    ev = torch.randn(5, H, W, device="cuda"); im = torch.randn(3, H, W, device="cuda")
    out = []
    for k, s in enumerate(enc.scales):
        st = conv_hip.MsState(H, W, s, "cuda")
        fn = lambda: conv_hip.ms_lstm_superstate_step(enc, k, ev, im, st, True, want_half=k > 0)
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(30):
            fn()
        b.record(); torch.cuda.synchronize()
        out.append("scale %d: %.1f us" % (s, a.elapsed_time(b) / 30 * 1e3))
    print("%dx%d  " % (W, H) + "  ".join(out), flush=True)
