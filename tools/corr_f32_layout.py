#!/usr/bin/env python
"""the fp32 tracker's correlation launch on its own graph, per mode of the patchifier's pack_f32 (= RAMP_CORR_F32_MFMA):
2 split fp16 pairs on the f16 matrix cores (corr_mfma_kernel<CorrX2>), 1 corr_mfma_kernel<float> on planes chunked as
[h][8][w][16], 0 plain NHWC planes (the same fp32 MFMA kernel here: slam.corr's fast path is asked for explicitly)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from rampvo_amd.config import make_cfg
from rampvo_amd.Ramp_vo import Ramp_vo
from rampvo_amd.synthetic import SyntheticStream, make_network
res = {}
for pack in [int(c) for c in os.environ.get("CORR_F32_MODES", "210210")]:
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
    if pack == 0:
        slam._f32_mode = 1                   # plain planes through corr_mfma_kernel<float, false> (mode 0 proper: the fmaf chain)
        out = slam.corr(coords, order=plan.g_ij.order)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(20):
        out = slam.corr(coords, order=plan.g_ij.order)
    e.record(); torch.cuda.synchronize()
    print("pack_f32=%s chunked=%s split=%s planes %s: E = %d, %.1f us per launch, checksum %.6f" % (
        pack, slam._chunked, slam._split, tuple(slam.fmap1_.shape), coords.shape[1], s.elapsed_time(e) * 1e3 / 20, float(out.double().sum())), flush=True)
