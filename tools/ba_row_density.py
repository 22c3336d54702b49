#!/usr/bin/env python
"""SURVEY 8(f) N1 (block-sparse E, ramp/fastba/block_e.cu): how sparse ARE the per-patch rows of E at the windows where the
reference's EfficentE would matter?  Tracks configs[2] / configs[4]-shaped snapshots on the GPU and counts, per patch, the
free poses its factors touch (a 6-wide block of its [6N] row each) -- the dense row's fill, and the Schur product's work in
both storages: dense split-K SYRK Mu (6N)^2 multiply-adds, block lookup sum_k (6 n_k)^2."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from rampvo_amd.config import make_cfg
from rampvo_amd.Ramp_vo import Ramp_vo
from rampvo_amd.synthetic import SyntheticStream, make_network

CASES = [("configs[2] (MultiScale 640x480, M=96, precise.yaml, every frame kept)", "MultiScale", "precise", 96, 480, 640, 56, dict(KEYFRAME_THRESH=0.0)),
         ("configs[4] (MultiScale 1280x720, M=256, window 32, precise lifetimes, every frame kept)", "MultiScale", "precise", 256, 720, 1280, 46, dict(KEYFRAME_THRESH=0.0, OPTIMIZATION_WINDOW=32)),
         ("configs[1] (SingleScale 640x480, M=96, default.yaml)", "SingleScale", "default", 96, 480, 640, 60, {})]
for name, mode, preset, M, H, W, T, over in CASES:
    cfg = make_cfg(preset, PATCHES_PER_FRAME=M, MIXED_PRECISION=True, **over)
    slam = Ramp_vo(cfg, make_network(mode), {"event_bias": True}, ht=H, wd=W)
    st = SyntheticStream(H, W, T + 1, seed=4321, device="cuda")
    with torch.no_grad():
        for t in range(T):
            im, ev, K, mask = st.frame(t)
            slam(t, input_tensor=(ev, im, mask), intrinsics=K)
    ii, jj, kk, n = slam._ii.copy(), slam._jj.copy(), slam._kk.copy(), slam.n
    N = min(cfg.OPTIMIZATION_WINDOW, n - 1)
    t0 = n - N
    free = lambda f: (f >= t0) & (f < n)
    keys = np.concatenate([kk[free(ii)] * 4096 + ii[free(ii)], kk[free(jj)] * 4096 + jj[free(jj)]])
    uk = np.unique(keys)
    patch, cnt = np.unique(uk // 4096, return_counts=True)
    Mu = len(np.unique(kk))
    dense = Mu * (6 * N) ** 2
    block = int(((6 * cnt.astype(np.int64)) ** 2).sum())
    print("%s\n  keyframes %d, factors %d, patches with factors %d (%d touch a free pose), free poses N = %d (6N = %d)"
          % (name, n, len(ii), Mu, len(patch), N, 6 * N))
    print("  blocks per row: mean %.1f of %d (fill %.2f), median %d, p10 %d, p90 %d" % (cnt.mean(), N, cnt.mean() / N, np.median(cnt), np.percentile(cnt, 10), np.percentile(cnt, 90)))
    print("  Schur product multiply-adds: dense rows %.1f M, block lookup %.1f M (%.2f x)  | E storage: dense %.1f MB, blocks %.1f MB"
          % (dense / 1e6, block / 1e6, block / dense, Mu * 6 * N * 4 / 1e6, cnt.sum() * 6 * 4 / 1e6))
    del slam
    torch.cuda.empty_cache()
