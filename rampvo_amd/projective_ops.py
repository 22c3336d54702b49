"""projective geometry of the tracking loop (reference: ramp/projective_ops.py).

Only the inference path is provided: ``transform`` without Jacobians (those feed
the training-time python BA), ``point_cloud`` and ``flow_mag``; each is one fused
HIP kernel instead of the reference's chain of ATen + lietorch launches."""
import torch

from . import ops
from .lietorch import SE3

MIN_DEPTH = 0.2


def _data(poses):
    return poses.data if isinstance(poses, SE3) else poses


def transform(poses, patches, intrinsics, ii, jj, kk, depth=False, valid=False, jacobian=False,
              tonly=False):
    """reference projective_ops.py:50-101 -> coords [1,E,P,P,2]"""
    if depth or valid or jacobian:
        raise NotImplementedError("inference path only (no depth/valid/jacobian outputs)")
    out = ops.transform(_data(poses), patches, intrinsics, ii, jj, kk, tonly)  # [1,E,2,P,P]
    return out.permute(0, 1, 3, 4, 2)


def reproject(poses, patches, intrinsics, ii, jj, kk, tonly=False):
    """Ramp_vo.reproject layout directly: [1,E,2,P,P] (ramp/Ramp_vo.py:184-192)"""
    return ops.transform(_data(poses), patches, intrinsics, ii, jj, kk, tonly)


def point_cloud(poses, patches, intrinsics, ix):
    """patch-centre 3-D points [m,3] (reference :103-105 followed by Ramp_vo.py:308-310)"""
    return ops.point_cloud(_data(poses), patches, intrinsics, ix)


def flow_mag(poses, patches, intrinsics, ii, jj, kk, beta=0.3):
    """reference projective_ops.py:108-118 -> [1,E,P,P]"""
    p = _data(poses)
    c0 = ops.transform(p, patches, intrinsics, ii, ii, kk, False)
    c1 = ops.transform(p, patches, intrinsics, ii, jj, kk, False)
    c2 = ops.transform(p, patches, intrinsics, ii, jj, kk, True)
    flow1 = (c1 - c0).norm(dim=2)
    flow2 = (c2 - c0).norm(dim=2)
    return beta * flow1 + (1 - beta) * flow2
