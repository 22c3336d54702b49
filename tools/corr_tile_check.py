#!/usr/bin/env python
"""coarse correlation level from LDS tiles (ramp_corr_fwd_tiled) against the gather kernel at the bench size: bit equality,
share of the factors the tiles take, time per call of each"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from rampvo_amd import ops
from rampvo_amd._lib import RAMP_NHWC8


def case(E=38600, M=96, slots=36, H=120, W=160, seed=0, window=22):
    g = torch.Generator(device="cuda").manual_seed(seed)
    rng = np.random.default_rng(seed)
    f1 = (torch.randn(slots * M, 3, 3, 128, generator=g, device="cuda") * 0.5).half()
    l1 = (torch.randn(slots, H, 16, W, 8, generator=g, device="cuda") * 0.5).half()
    l4 = (torch.randn(slots, H // 4, 16, W // 4, 8, generator=g, device="cuda") * 0.5).half()
    # factors: target-frame major, a patch centre anywhere in (and a little around) the image, pixels 1 px apart
    jj = np.sort(rng.integers(100, 100 + window, E)).astype(np.int64)
    kk = rng.integers(0, 3 * slots * M, E).astype(np.int64)
    cx = rng.uniform(-12, W + 12, E); cy = rng.uniform(-12, H + 12, E)
    sc = rng.uniform(0.7, 1.5, E)
    sc[rng.random(E) < 0.01] = 9.0            # spread-out patches: ragged path / too wide for a tile
    d = np.arange(-1, 2, dtype=np.float64)
    x = cx[:, None, None] + sc[:, None, None] * d[None, None, :] + rng.normal(0, 0.05, (E, 3, 3))
    y = cy[:, None, None] + sc[:, None, None] * d[None, :, None] + rng.normal(0, 0.05, (E, 3, 3))
    coords = torch.from_numpy(np.stack([x, y], 1).astype(np.float32)).cuda()
    far = rng.random(E) < float(os.environ.get("CT_FAR", 0.45))
    coords[torch.from_numpy(far).cuda()] += 1000.0     # nothing in the plane
    return f1, l1, l4, coords, torch.from_numpy(kk).cuda(), torch.from_numpy(jj).cuda(), M, slots


def timed(fn, n=20):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


if __name__ == "__main__":
    f1, l1, l4, coords, kk, jj, M, slots = case()
    E = coords.shape[0]
    order = torch.arange(E, dtype=torch.int32, device="cuda")
    ws = ops.corr_tile_workspace(E, slots, l4.shape[1], l4.shape[3])
    kw = dict(order=order, row_elems=896, mod_ii=slots * M, mod_jj=slots)
    base = ops.corr(f1, [l1, l4], coords, kk, jj, 3, (1.0, 4.0), RAMP_NHWC8, **kw)
    for rep in range(0 if os.environ.get("CT_QUICK") else 3):
        out = ops.corr(f1, [l1, l4], coords, kk, jj, 3, (1.0, 4.0), RAMP_NHWC8, tile_ws=ws, **kw)
        same = torch.equal(torch.nan_to_num(out.float(), nan=-7.0), torch.nan_to_num(base.float(), nan=-7.0))
        nbins = slots * ((l4.shape[3] + 16) // 10) * ((l4.shape[1] + 16) // 10)
        hdr = ws[:4 * (16 + nbins)].view(torch.int32)
        print("call", rep, "bit-identical:", same, "| counters left:", int(hdr.abs().sum()))
        if not same:
            dif = (out.float() - base.float()).abs()
            bad = (dif > 0).nonzero()
            print("  mismatches:", bad.shape[0], "first:", bad[:5].tolist(), "max", float(dif.max()))
    if os.environ.get("CT_QUICK"):          # kernel times from HIP events around single kernels are not available: whole call
        t_t = timed(lambda: ops.corr(f1, [l1, l4], coords, kk, jj, 3, (1.0, 4.0), RAMP_NHWC8, tile_ws=ws, **kw))
        print("bin + tiles + list: %.1f us" % t_t)
        sys.exit(0)
    t_g = timed(lambda: ops.corr(f1, [l1, l4], coords, kk, jj, 3, (1.0, 4.0), RAMP_NHWC8, **kw))
    t_t = timed(lambda: ops.corr(f1, [l1, l4], coords, kk, jj, 3, (1.0, 4.0), RAMP_NHWC8, tile_ws=ws, **kw))
    print("gather kernel, both levels: %.1f us | bin + tiles + gather (fine level): %.1f us" % (t_g, t_t))


def window_stats(coords, H, W, cdv):
    """union-window sizes of the factors at one level (numpy restatement of corr_geom)"""
    c = coords.detach().cpu().numpy().astype(np.float32) / np.float32(cdv)
    fx = np.floor(c[:, 0].reshape(-1, 9)).astype(np.int64); fy = np.floor(c[:, 1].reshape(-1, 9)).astype(np.int64)
    live = (fx - 3 < W) & (fx - 3 + 8 > 0) & (fy - 3 < H) & (fy - 3 + 8 > 0)
    big = 1 << 30
    minx = np.where(live, fx, big).min(1); maxx = np.where(live, fx, -big).max(1)
    miny = np.where(live, fy, big).min(1); maxy = np.where(live, fy, -big).max(1)
    n = live.sum(1)
    bw = np.where(n > 0, maxx - minx + 8, 0); bh = np.where(n > 0, maxy - miny + 8, 0)
    return n, bw, bh
