#!/usr/bin/env python
"""does the front-end graph write anything it does not own (and vice versa)?  Sequential tracker; at a few frames:
checksum everything the update/BA side owns, replay the front end, compare; checksum everything the front end owns,
run update() + keyframe(), compare."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from rampvo_amd import _lib
from rampvo_amd.config import make_cfg
from rampvo_amd.Ramp_vo import Ramp_vo
from rampvo_amd.synthetic import SyntheticStream, make_network
net = make_network("SingleScale")
slam = Ramp_vo(make_cfg("default", PATCHES_PER_FRAME=96, MIXED_PRECISION=True), net, {"event_bias": True})
T = 120
stream = SyntheticStream(480, 640, T + 1, seed=100, device="cuda")
frames = [stream.frame(t) for t in range(T)]
def bits(t):
    return t.contiguous().view(torch.uint8).to(torch.int64).sum().item() if t.numel() else 0
def main_state():
    d = {k: getattr(slam, k) for k in ("poses_", "patches_", "intrinsics_", "points_", "colors_", "tstamps_", "index_",
                                          "index_map_", "imap_", "gmap_", "fmap1_", "fmap2_", "_net_buf", "ii", "jj", "kk")}
    if slam._plan is not None:
        for i, t in enumerate(slam._plan.tensors()):
            d["plan%d" % i] = t
    for k, v in _lib._ws_cache.items():
        d["ws_%s" % (k[-1],) + str(k[1])] = v
    for i, (n, p) in enumerate(net.update.named_parameters()):
        d["upd_" + n] = p
    fu = net.update.fused(torch.float16)
    for k, v in fu.weights().items():
        for j, t in enumerate(v if isinstance(v, (tuple, list)) else [v]):
            if isinstance(t, torch.Tensor):
                d["w_%s_%d" % (k, j)] = t
            elif isinstance(t, (list, tuple)):
                for jj_, u in enumerate(t):
                    if isinstance(u, torch.Tensor):
                        d["w_%s_%d_%d" % (k, j, jj_)] = u
    return d
def fe_state():
    pat = net.patchify
    d = {}
    st = pat.encoder._hip_state
    for k in getattr(st, "__slots__", ()):
        v = getattr(st, k)
        if isinstance(v, torch.Tensor):
            d["lstm_" + k] = v
    for key, g in pat._graphs.items():
        graph, ev_s, im_s, outs, extra, sig = g
        d["ev_s"], d["im_s"] = ev_s, im_s
        for i, o in enumerate(outs):
            if isinstance(o, torch.Tensor):
                d["out%d" % i] = o
        for k, v in (extra or {}).items():
            if isinstance(v, torch.Tensor):
                d["extra_" + k] = v
    for n, p in pat.named_parameters():
        d["enc_" + n] = p
    return d
with torch.no_grad():
    for t in range(T):
        im, ev, K, mask = frames[t]
        if t >= 40 and t % 10 == 0:
            torch.cuda.synchronize()
            a = {k: bits(v) for k, v in main_state().items()}
            net.patchify(input_=(ev, im, mask), patches_per_image=96, event_bias=True, reinit_hidden=False)   # extra FE replay
            torch.cuda.synchronize()
            b = {k: bits(v) for k, v in main_state().items()}
            ch = [k for k in a if a[k] != b.get(k)]
            print("t=%d  front-end replay changed update-side tensors: %s" % (t, ch or "none"), flush=True)
        slam(t, input_tensor=(ev, im, mask), intrinsics=K)
        if t >= 40 and t % 10 == 5:
            torch.cuda.synchronize()
            a = {k: bits(v) for k, v in fe_state().items()}
            slam.update(); slam.keyframe()
            torch.cuda.synchronize()
            b = {k: bits(v) for k, v in fe_state().items()}
            ch = [k for k in a if a[k] != b.get(k)]
            print("t=%d  update()+keyframe() changed front-end tensors: %s" % (t, ch or "none"), flush=True)
