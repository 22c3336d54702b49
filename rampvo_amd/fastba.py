"""fastba operator surface (reference: ramp/fastba/ba.py:1-8, ba.cpp:183-189)."""
import torch

from . import ops


def BA(poses, patches, intrinsics, target, weight, lmbda, ii, jj, kk, t0, t1, M=None,
       iterations=2, eff_impl=False, info=None, plan=None):
    """cuda_ba.forward: in-place GN bundle adjustment.  ``M`` (patches per
    frame) and ``eff_impl`` only select the reference's block-sparse E storage;
    this implementation never materialises the dense [6N x Mu] E either way."""
    p = poses.data if hasattr(poses, "data") and not isinstance(poses, torch.Tensor) else poses
    if plan is not None:
        ops.ba(p, patches, intrinsics, target, weight, lmbda, ii, jj, kk, t0, t1, iterations, info, plan=plan)
    else:
        ops.ba(p, patches, intrinsics, target, weight, lmbda, ii, jj, kk, t0, t1, iterations, info)
    return []


def neighbors(ii, jj, ii_bound=0, jj_bound=0):
    """cuda_ba.neighbors(kk, jj) -> (ix, jx), computed on the device."""
    return ops.neighbors(ii, jj, ii_bound, jj_bound)


def reproject(poses, patches, intrinsics, ii, jj, kk):
    return ops.reproject(poses, patches, intrinsics, ii, jj, kk)
