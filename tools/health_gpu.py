#!/usr/bin/env python
"""Numerical health of a long tracker run on the GPU (the bench workload): every `--every` frames print n, E, finiteness
and magnitudes of poses / depths / hidden state / features.  usage: tools/health_gpu.py [--mixed 1] [--frames 900]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rampvo_amd.config import make_cfg                      # noqa: E402
from rampvo_amd.Ramp_vo import Ramp_vo                      # noqa: E402
from rampvo_amd.synthetic import SyntheticStream, make_network   # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--mixed", type=int, default=1)
ap.add_argument("--frames", type=int, default=900)
ap.add_argument("--every", type=int, default=50)
ap.add_argument("--mode", default="SingleScale")
ap.add_argument("--profile", default="wide")
ap.add_argument("--pipelined", type=int, default=0)
ap.add_argument("--w-bias", type=float, default=None)
ap.add_argument("--seed", type=int, default=1234)
a = ap.parse_args()
net = make_network(a.mode, profile=a.profile, w_bias=a.w_bias)
slam = Ramp_vo(make_cfg("default", PATCHES_PER_FRAME=96, MIXED_PRECISION=bool(a.mixed)), net, {"event_bias": True})
slam.inputs_ready = bool(a.pipelined)
stream = SyntheticStream(480, 640, a.frames + 1, seed=a.seed, device="cuda")
frames = [stream.frame(t) for t in range(a.frames)]
torch.cuda.synchronize()
with torch.no_grad():
    for t in range(a.frames):
        im, ev, K, mask = frames[t]
        slam(t, input_tensor=(ev, im, mask), intrinsics=K)
        if t % a.every == a.every - 1:
            slam.settle()
            n = slam.n
            p, d = slam.poses_[:n], slam.patches_[:n, :, 2]
            netb = slam._net_buf
            w = getattr(slam, "last_weight", None)
            print("t=%d n=%d E=%d poses finite=%s |t|max=%.3g depth[%.3g,%.3g] finite=%s | net finite=%s max=%.3g | "
                  "fmap max=%.3g gmap max=%.3g imap max=%.3g | weight nan=%s" % (
                      t, n, len(slam._ii), bool(torch.isfinite(p).all()), float(p[:, :3].abs().nan_to_num(0).max()),
                      float(d.nan_to_num(0).min()), float(d.nan_to_num(0).max()), bool(torch.isfinite(d).all()),
                      bool(torch.isfinite(netb).all()), float(netb.abs().nan_to_num(0).max()),
                      float(slam.fmap1_.float().abs().nan_to_num(0).max()), float(slam.gmap_.float().abs().nan_to_num(0).max()),
                      float(slam.imap_.float().abs().nan_to_num(0).max()),
                      bool(torch.isnan(w).any()) if w is not None else None), flush=True)
