"""Seeded synthetic problem instances shared by the CPU (oracle) and GPU (parity) tests."""
import numpy as np


def se3_exp_np(xi):
    """tiny numpy SE3 exp (tests only) -> [tx ty tz qx qy qz qw]"""
    import oracle as orc
    return orc.se3_exp(np.asarray(xi, np.float32))


def ba_scene(seed=0, n_frames=8, M=12, lifetime=5, H=120, W=160, noise=1.5, n_total_frames=None,
             far=False):
    """A small sliding-window graph shaped like Ramp_vo's: every patch of frame i is
    connected to the frames within `lifetime` of i (both directions, incl. i itself)."""
    rng = np.random.default_rng(seed)
    N = n_total_frames or n_frames
    xi = np.zeros((N, 6), np.float32)
    xi[:n_frames] = np.cumsum(rng.normal(0, [0.05, 0.03, 0.02, 0.01, 0.01, 0.01], (n_frames, 6)), 0)
    import oracle as orc
    poses = orc.se3_exp(xi).astype(np.float32)
    poses[n_frames:] = [0, 0, 0, 0, 0, 0, 1]
    fx = fy = W * 0.5
    cx, cy = W * 0.5, H * 0.5
    intr = np.tile(np.array([fx, fy, cx, cy], np.float32), (N, 1))
    patches = np.zeros((N * M, 3, 3, 3), np.float32)
    for f in range(n_frames):
        for m in range(M):
            x = rng.uniform(8, W - 8) + rng.uniform(0, 1)
            y = rng.uniform(8, H - 8)
            gx, gy = np.meshgrid(np.arange(-1, 2), np.arange(-1, 2))
            patches[f * M + m, 0] = x + gx
            patches[f * M + m, 1] = y + gy
            patches[f * M + m, 2] = rng.uniform(0.15, 1.2)
    ii, jj, kk = [], [], []
    for f in range(n_frames):
        for m in range(M):
            for j in range(max(0, f - lifetime), min(n_frames, f + lifetime + 1)):
                ii.append(f); jj.append(j); kk.append(f * M + m)
    ii, jj, kk = (np.asarray(a, np.int64) for a in (ii, jj, kk))
    perm = rng.permutation(len(ii)) if far else np.arange(len(ii))
    ii, jj, kk = ii[perm], jj[perm], kk[perm]
    coords = orc.transform(poses, patches, intr, ii, jj, kk)[0]          # [E,2,3,3]
    target = coords[:, :, 1, 1] + rng.normal(0, noise, (len(ii), 2)).astype(np.float32)
    weight = rng.uniform(0.0, 1.0, (len(ii), 2)).astype(np.float32)
    # a few gross outliers / out-of-image targets exercise the validity mask
    bad = rng.choice(len(ii), size=max(1, len(ii) // 40), replace=False)
    target[bad] += rng.normal(0, 300, (len(bad), 2)).astype(np.float32)
    return dict(poses=poses, patches=patches, intr=intr, ii=ii, jj=jj, kk=kk,
                target=target.astype(np.float32), weight=weight,
                lmbda=np.array([1e-4], np.float32), n_frames=n_frames, M=M)


def ba_pin_scene(seed, n_frames, M=8, lifetime=4, H=120, W=160, noise=0.4):
    """float64 problem for the fp64 pin of the bundle-adjustment algebra (tests/golden/ba_f64_pin.npz): ba_scene's graph and
    geometry without anything that makes the two implementations under comparison -- ramp/ba.py::BA and the cuda_ba
    restatement -- take different branches: no outliers (residual gates 250 vs 128 px), every depth in
    (0.2, 1.2) and every point in front of every camera (Z > 0.2), projections inside the bounds, one intrinsics row"""
    import oracle as orc
    rng = np.random.default_rng(seed)
    xi = np.cumsum(rng.normal(0, [0.03, 0.02, 0.01, 0.004, 0.004, 0.004], (n_frames, 6)), 0)
    poses = orc.se3_exp_f64(xi)
    intr = np.tile(np.array([W * 0.5, W * 0.5, W * 0.5, H * 0.5], np.float64), (n_frames, 1))
    patches = np.zeros((n_frames * M, 3, 3, 3), np.float64)
    gx, gy = np.meshgrid(np.arange(-1, 2), np.arange(-1, 2))
    for k in range(n_frames * M):
        patches[k, 0] = rng.uniform(20, W - 20) + gx
        patches[k, 1] = rng.uniform(20, H - 20) + gy
        patches[k, 2] = rng.uniform(0.2, 1.2)
    ii, jj, kk = [], [], []
    for f in range(n_frames):
        for m in range(M):
            for j in range(max(0, f - lifetime), min(n_frames, f + lifetime + 1)):
                ii.append(f); jj.append(j); kk.append(f * M + m)
    ii, jj, kk = (np.asarray(a, np.int64) for a in (ii, jj, kk))
    perm = rng.permutation(len(ii))
    ii, jj, kk = ii[perm], jj[perm], kk[perm]
    xy = orc.ba_edge_terms(poses, patches, intr, ii, jj, kk, f64=True)["xy"]
    target = xy + rng.normal(0, noise, xy.shape)
    weight = rng.uniform(0.05, 1.0, xy.shape)
    return dict(poses=poses, patches=patches, intr=intr, ii=ii, jj=jj, kk=kk, target=target, weight=weight,
                lmbda=np.array([1e-4]), n_frames=n_frames, M=M)


def corr_case(seed=0, E=48, N1=40, N2=6, H=30, W=40, C=128, distort=True, wide=False):
    rng = np.random.default_rng(seed)
    fmap1 = rng.normal(0, 1, (1, N1, C, 3, 3)).astype(np.float32)
    fmap2 = rng.normal(0, 1, (1, N2, C, H, W)).astype(np.float32)
    cx = rng.uniform(-6, W + 6, (E, 1, 1, 1)).astype(np.float32)
    cy = rng.uniform(-6, H + 6, (E, 1, 1, 1)).astype(np.float32)
    gy, gx = np.meshgrid(np.arange(-1, 2), np.arange(-1, 2), indexing="ij")
    scale = rng.uniform(0.6, 1.6, (E, 1, 1)).astype(np.float32)
    if distort:
        scale[::7] *= 6.0          # union of windows > 128 px -> per-pixel fallback path
    if wide:
        scale[1::3] = rng.uniform(1.7, 2.4, scale[1::3].shape)     # 12 x 12 .. 13 x 14 unions (one pass up to 192 px)
    x = cx[:, 0] + gx[None] * scale + rng.normal(0, 0.05, (E, 3, 3))
    y = cy[:, 0] + gy[None] * scale + rng.normal(0, 0.05, (E, 3, 3))
    coords = np.stack([x, y], 1).astype(np.float32)[None]          # [1,E,2,3,3]
    coords[0, 3] = 1e9            # absurd coordinates: every window off-image
    coords[0, 5, :, 0, 0] = -40.0  # one dead pixel inside an otherwise live patch
    ii = rng.integers(0, N1, E).astype(np.int64)
    jj = rng.integers(0, N2, E).astype(np.int64)
    return fmap1, fmap2, coords, ii, jj
