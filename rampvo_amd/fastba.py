"""fastba operator surface (reference: ramp/fastba/ba.py:1-8, ba.cpp:183-189)."""
import torch

from . import ops


def BA(poses, patches, intrinsics, target, weight, lmbda, ii, jj, kk, t0, t1, M=None,
       iterations=2, eff_impl=False, info=None, plan=None):
    """cuda_ba.forward: in-place GN bundle adjustment (reference ba_cuda.cu:433-582).

    ``M`` (patches per frame, the reference's PPF) and ``eff_impl`` select the reference's storage of the pose-depth
    coupling E: a dense [6N x Mu] matrix, or ``EfficentE``'s per-(i, j)-block lookup (fastba/block_e.cu).  Both are the
    same algebra (oracle: ``orc.ba(eff_ppf=M)`` restates the lookup kernels and agrees with the dense restatement to
    rounding).  This implementation has ONE storage and serves both settings with it: one row of 6N values per
    patch that actually occurs in the graph (``Erow [Mu][6N]``, 8.3 MB at configs[4]: Mu = 10,752, 6N = 192), formed
    by an ordered segment sum over the patch's edges and consumed by a split-K SYRK -- neither the dense matrix's
    atomic scatter nor the lookup's (frames x frames) index tensor exists here, so there is nothing for the flag to
    switch.  tests/test_ops_gpu.py::test_ba_eff_impl_at_config5_size pins it against the oracle's lookup path."""
    p = poses.data if hasattr(poses, "data") and not isinstance(poses, torch.Tensor) else poses
    if eff_impl and not M:
        raise ValueError("eff_impl=True needs M (patches per frame), as cuda_ba.forward's PPF")
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
