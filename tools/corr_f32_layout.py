#!/usr/bin/env python
"""corr_mfma_kernel<float> on the tracker's own graph: planes chunked as [h][8][w][16] (round 6) against plain NHWC planes
(same values, same kernel: the layout only changes which bytes a load instruction fetches together)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from rampvo_amd.config import make_cfg
from rampvo_amd.Ramp_vo import Ramp_vo
from rampvo_amd.synthetic import SyntheticStream, make_network
res = {}
for pack in (True, False, True, False):
    torch.manual_seed(1234)
    net = make_network("SingleScale")
    net.patchify.pack_f32 = pack
    slam = Ramp_vo(make_cfg("default", PATCHES_PER_FRAME=96, MIXED_PRECISION=False), net, {"event_bias": True})
    slam.device_steps = False
    st = SyntheticStream(480, 640, 60, seed=1234, device="cuda")
    for t in range(60):
        im, ev, K, m = st.frame(t)
        slam(t, input_tensor=(ev, im, m), intrinsics=K)
    plan = slam._graph_plan()
    coords = slam.reproject()
    for _ in range(3):
        out = slam.corr(coords, order=plan.g_ij.order)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(20):
        out = slam.corr(coords, order=plan.g_ij.order)
    e.record(); torch.cuda.synchronize()
    print("pack_f32=%s chunked=%s planes %s: E = %d, %.1f us per launch, checksum %.6f" % (
        pack, slam._chunked, tuple(slam.fmap1_.shape), coords.shape[1], s.elapsed_time(e) * 1e3 / 20, float(out.double().sum())), flush=True)
