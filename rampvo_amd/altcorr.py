"""altcorr operator surface (reference: ramp/altcorr/correlation.py:51-72).

Forward only: the reference's autograd wrappers / backward kernels serve
training, which is outside the tracking hot path.
"""
import torch

from . import ops
from ._lib import RAMP_NCHW, RAMP_NHWC


def patchify(net, coords, radius, mode='bilinear'):
    """extract (2r+1)^2 patches around ``coords`` (reference correlation.py:51-68).

    net [n,C,H,W], coords [n,M,2] -> [n,M,C,2r+1,2r+1]; mode other than
    'bilinear' returns the raw (2r+2)^2 windows of cuda_corr.patchify_forward."""
    return ops.patchify(net, coords, radius, bilinear=(mode == 'bilinear'))


def corr(fmap1, fmap2, coords, ii, jj, radius=1, dropout=1):
    """reference correlation.py:71-72 / cuda_corr.forward.

    fmap1 [1,N1,C,P,P], fmap2 [1,N2,C,H,W], coords [1,E,2,P,P] ->
    [1,E,2r+1,2r+1,P,P] (x-offset axis first, contiguous).  ``dropout`` only
    affects the reference's backward pass and is ignored."""
    assert fmap1.shape[0] == 1 and fmap2.shape[0] == 1 and coords.shape[0] == 1, "batch 1 only"
    out = ops.corr(fmap1[0], [fmap2[0]], coords[0], ii, jj, radius, (1.0,), RAMP_NCHW)
    return out[..., 0].unsqueeze(0)


def corr_pyramid(gmap, pyramid, coords, ii, jj, radius=3, levels=(1, 4), layout=RAMP_NCHW, order=None,
                 row_elems=0, mod_ii=0, mod_jj=0, fast_f32=None):
    """fused form of Ramp_vo.corr (ramp/Ramp_vo.py:175-182): all levels in one
    launch, result already stacked as [1, E, (2r+1)^2 * P^2 * nlevels]."""
    out = ops.corr(gmap, list(pyramid), coords, ii, jj, radius, tuple(float(l) for l in levels),
                   layout, order=order, row_elems=row_elems, mod_ii=mod_ii, mod_jj=mod_jj, fast_f32=fast_f32)
    return out.view(1, out.shape[0], -1)
