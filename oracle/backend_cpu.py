"""CPU backend for the host logic of rampvo_amd -- TEST / CPU-BASELINE ONLY.

``rampvo_amd`` has no CPU path: its operators call HIP kernels or raise.  To
exercise the *host* logic (graph bookkeeping, Ramp_vo state machine, network
glue) without a GPU, and to time the "port" CPU baseline in ``bench.py``, this
module swaps the functions of ``rampvo_amd.ops`` for oracle-backed CPU versions
inside a context manager.  It lives under ``oracle/`` and is imported only by
``tests/`` and the ``cpu_baseline`` leg of ``bench.py``.
"""
import contextlib

import numpy as np
import torch

import oracle as orc

RAMP_NCHW, RAMP_NHWC = 0, 1


def _np(t):
    return t.detach().cpu().float().contiguous().numpy()


def _t(a, like=None):
    out = torch.from_numpy(np.ascontiguousarray(a))
    return out.to(like.dtype) if like is not None else out


def patchify(net, coords, radius, bilinear=True, layout=RAMP_NCHW, out_layout=RAMP_NCHW):
    x = net if layout == RAMP_NCHW else net.permute(0, 3, 1, 2)
    fn = orc.patchify if bilinear else orc.patchify_raw
    out = _t(fn(_np(x), _np(coords), radius), net)
    return out if out_layout == RAMP_NCHW else out.permute(0, 1, 3, 4, 2).contiguous()


def corr(fmap1, fmaps2, coords, ii, jj, radius=3, coord_divs=(1.0,), layout=RAMP_NCHW, order=None, row_elems=0, fast_f32=None, mod_ii=0, mod_jj=0):
    f1 = fmap1 if layout == RAMP_NCHW else fmap1.permute(0, 3, 1, 2)
    if mod_ii:                        # the tracker's ring-buffer slots (reference Ramp_vo.py:178-179)
        ii = ii % int(mod_ii)
    if mod_jj:
        jj = jj % int(mod_jj)
    outs = []
    for f2, dv in zip(fmaps2, coord_divs):
        f2 = f2 if layout == RAMP_NCHW else f2.permute(0, 3, 1, 2)
        c = (coords.float() / dv)
        outs.append(orc.corr(_np(f1)[None], _np(f2)[None], _np(c)[None], ii.numpy(), jj.numpy(), radius)[0])
    return _t(np.stack(outs, -1), fmap1)


def se3_unary(name, x, din, dout):
    return _t(getattr(orc, name.replace("ramp_", ""))(_np(x)))


def se3_binary(name, x, y, dx, dy, dout):
    return _t(getattr(orc, name.replace("ramp_", ""))(_np(x), _np(y)))


def transform(poses, patches, intrinsics, ii, jj, kk, tonly=False):
    return _t(orc.transform(_np(poses), _np(patches), _np(intrinsics), ii.numpy(), jj.numpy(), kk.numpy(), tonly))


def reproject(poses, patches, intrinsics, ii, jj, kk):
    return _t(orc.reproject(_np(poses), _np(patches), _np(intrinsics), ii.numpy(), jj.numpy(), kk.numpy()))


def point_cloud(poses, patches, intrinsics, ix):
    P = patches.shape[-1]
    pt = _np(patches).reshape(-1, 3, P, P)
    ixn = ix.numpy()
    m = len(ixn)
    K = _np(intrinsics).reshape(-1, 4)[ixn]
    c = pt[:m, :, P // 2, P // 2]
    X0 = np.stack([(c[:, 0] - K[:, 2]) / K[:, 0], (c[:, 1] - K[:, 3]) / K[:, 1], np.ones(m, np.float32), c[:, 2]], -1)
    Tinv = orc.se3_inv(_np(poses).reshape(-1, 7)[ixn])
    Pw = orc.se3_act4(Tinv, X0.astype(np.float32))
    return _t(Pw[:, :3] / Pw[:, 3:])


class Groups:
    pass


def group_by(keys, key_bound=0):
    k = keys.numpy()
    g = Groups()
    g.E = len(k)
    uk, inv = np.unique(k, return_inverse=True)
    order = np.argsort(k, kind="stable")
    seg = np.concatenate([[0], np.cumsum(np.bincount(inv, minlength=len(uk)))]) if len(k) else np.zeros(1)
    g.order = torch.from_numpy(order.astype(np.int32))
    g.gid = torch.from_numpy(inv.astype(np.int32))
    g.seg_start = torch.from_numpy(seg.astype(np.int32))
    g.ukeys = torch.from_numpy(uk.astype(np.int64))
    g.ngroups = torch.tensor([len(uk)], dtype=torch.int32)
    return g


def neighbors(kk, jj, kk_bound=0, jj_bound=0):
    ix, jx = orc.neighbors(kk.numpy(), jj.numpy())
    return torch.from_numpy(ix), torch.from_numpy(jx)


def segment_softmax_sum(fx, gx, groups, max_groups):
    G = int(groups.ngroups.item())
    y = orc.segment_softmax_sum(_np(fx), _np(gx), groups.gid.numpy().astype(np.int64), G)
    out = torch.zeros((max_groups, fx.shape[1]), dtype=fx.dtype)
    out[:G] = _t(y, fx)
    return out


def ba(poses, patches, intrinsics, target, weight, lmbda, ii, jj, kk, t0, t1, iterations=2, info=None):
    assert poses.is_contiguous() and patches.is_contiguous() and poses.dtype == torch.float32
    P = patches.shape[-1]
    st = orc.ba(poses.view(-1, 7).numpy(), patches.view(-1, 3, P, P).numpy(), _np(intrinsics), _np(target),
                _np(weight), _np(lmbda), ii.numpy(), jj.numpy(), kk.numpy(), t0, t1, iterations)
    if info is not None:
        info.fill_(st)


_NAMES = ("patchify", "corr", "se3_unary", "se3_binary", "transform", "reproject", "point_cloud", "group_by",
          "neighbors", "segment_softmax_sum", "ba")


@contextlib.contextmanager
def cpu_oracle_ops(fp16_policy=False):
    """temporarily serve rampvo_amd.ops from the CPU oracle and the network modules' forwards from their plain-PyTorch
    restatements (oracle/host_cpu.py) -- tests / cpu_baseline only.  fp16_policy: the update operator with the shipped
    MIXED_PRECISION path's rounding points (fp16 Linear operands / outputs, fp32 accumulation and row arithmetic)"""
    import rampvo_amd.ops as ops
    from oracle import host_cpu
    saved = {n: getattr(ops, n) for n in _NAMES}
    patches = host_cpu.module_patches(fp16_policy)
    saved_mod = [(c, a, c.__dict__[a]) for c, a, _ in patches]
    try:
        for n in _NAMES:
            setattr(ops, n, globals()[n])
        for c, a, f in patches:
            setattr(c, a, f)
        yield
    finally:
        for n, f in saved.items():
            setattr(ops, n, f)
        for c, a, f in saved_mod:
            setattr(c, a, f)


def Ramp_vo(cfg, network, train_cfg, ht=480, wd=640, device="cpu"):
    """the oracle-side tracker (rampvo_amd.Ramp_vo with its device steps served by the oracle)"""
    from oracle.host_cpu import RampVoCPU
    return RampVoCPU(cfg, network, train_cfg, ht=ht, wd=wd, device=device)
