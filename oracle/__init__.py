"""CPU oracle for the RAMP-VO hot path -- TEST INFRASTRUCTURE ONLY.

numpy/ctypes front-end of ``oracle/ramp_oracle.c`` (a plain-C restatement of
the reference's native operators, each function citing the reference file:line
it follows).  Only ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` may import this package; nothing under
``rampvo_amd/`` does.

Parity status: lietorch ops are pinned by the identities of the reference's
``ramp/lietorch/run_tests.py``; altcorr / fastba have no upstream golden
vectors ("parity unpinned") and are anchored on the reference's own python
call sites executed in the build container (``oracle/make_golden.py``) plus
the ``ramp/ba.py`` cross-check.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libramp_oracle.so")
_lib = None


def build(force=False):
    """Compile libramp_oracle.so with gcc (recipe: oracle/Makefile)."""
    src = os.path.join(_HERE, "ramp_oracle.c")
    if (not force and os.path.exists(_LIB_PATH)
            and os.path.getmtime(_LIB_PATH) >= os.path.getmtime(src)):
        return _LIB_PATH
    subprocess.check_call(["make", "-s", "-C", _HERE, "-B", "all"])
    return _LIB_PATH


_lib64 = None


def lib64():
    """fp64 diagnostic build of the same source (rounding-envelope measurements only)"""
    global _lib64
    if _lib64 is None:
        path = os.path.join(_HERE, "libramp_oracle_f64.so")
        src = os.path.join(_HERE, "ramp_oracle.c")
        if not os.path.exists(path) or os.path.getmtime(path) < os.path.getmtime(src):
            subprocess.check_call(["make", "-s", "-C", _HERE, "libramp_oracle_f64.so"])
        _lib64 = ctypes.CDLL(path)
        _lib64.orc_ba.restype = ctypes.c_int
    return _lib64


def ba_f64(poses, patches, intrinsics, target, weight, lmbda, ii, jj, kk, t0, t1, iterations=2):
    """fp64 evaluation of orc_ba; returns (poses, patches) as float64 copies"""
    d = lambda a: np.ascontiguousarray(a, dtype=np.float64)
    P = patches.shape[-1]
    p, pt = d(poses).reshape(-1, 7).copy(), d(patches).reshape(-1, 3, P, P).copy()
    intrinsics, target, weight, lmbda = d(intrinsics).reshape(-1, 4), d(target).reshape(-1, 2), d(weight).reshape(-1, 2), d(lmbda).reshape(-1)
    ii, jj, kk = _i(ii), _i(jj), _i(kk)
    lib64().orc_ba(_p(p), _p(pt), _p(intrinsics), _p(target), _p(weight), _p(lmbda), _p(ii), _p(jj),
                   _p(kk), ii.shape[0], P, int(t0), int(t1), int(iterations))
    return p, pt


def _un64(name, x, din, dout):
    x = np.ascontiguousarray(x, dtype=np.float64)
    shp = x.shape[:-1]
    x2 = np.ascontiguousarray(x.reshape(-1, din))
    out = np.empty((x2.shape[0], dout), np.float64)
    getattr(lib64(), name)(_p(x2), _p(out), x2.shape[0])
    return out.reshape(shp + (dout,))


def _bin64(name, x, y, dx, dy, dout):
    x, y = np.asarray(x, np.float64), np.asarray(y, np.float64)
    bs = np.broadcast_shapes(x.shape[:-1], y.shape[:-1])
    x2 = np.ascontiguousarray(np.broadcast_to(x, bs + (dx,)).reshape(-1, dx))
    y2 = np.ascontiguousarray(np.broadcast_to(y, bs + (dy,)).reshape(-1, dy))
    out = np.empty((x2.shape[0], dout), np.float64)
    getattr(lib64(), name)(_p(x2), _p(y2), _p(out), x2.shape[0])
    return out.reshape(bs + (dout,))


# the SE3 forward ops of the fp64 build (oracle/refharness.py serves float64 tensors of the reference's python with them:
# ramp/ba.py + ramp/projective_ops.py in float64 are the independent pin of the bundle-adjustment restatement)
def se3_exp_f64(a): return _un64("orc_se3_exp", a, 6, 7)
def se3_log_f64(X): return _un64("orc_se3_log", X, 7, 6)
def se3_inv_f64(X): return _un64("orc_se3_inv", X, 7, 7)
def se3_mul_f64(X, Y): return _bin64("orc_se3_mul", X, Y, 7, 7, 7)
def se3_act4_f64(X, p): return _bin64("orc_se3_act4", X, p, 7, 4, 4)
def se3_adj_f64(X, a): return _bin64("orc_se3_adj", X, a, 7, 6, 6)
def se3_adjT_f64(X, a): return _bin64("orc_se3_adjT", X, a, 7, 6, 6)


def ba_edge_terms(poses, patches, intrinsics, ii, jj, kk, f64=False):
    """per-factor terms of the bundle-adjustment kernel (ba_cuda.cu:232-376 as restated in ramp_oracle.c::ba_edge_terms):
    returns dict(Ji [E,2,6] -- the kernel's sign: d/d(pose i) = -Ji --, Jj [E,2,6], Jz [E,2], xy [E,2] projected centres)"""
    T = np.float64 if f64 else np.float32
    L = lib64() if f64 else lib()
    c = lambda a: np.ascontiguousarray(a, dtype=T)
    P = patches.shape[-1]
    poses, patches, intrinsics = c(poses).reshape(-1, 7), c(patches).reshape(-1, 3, P, P), c(intrinsics).reshape(-1, 4)
    ii, jj, kk = _i(ii), _i(jj), _i(kk)
    E = ii.shape[0]
    Ji, Jj, Jz = np.empty((E, 2, 6), T), np.empty((E, 2, 6), T), np.empty((E, 2), T)
    xy, mask = np.empty((E, 2), T), np.empty(E, T)
    L.orc_ba_edge_terms(_p(poses), _p(patches), _p(intrinsics), _p(ii), _p(jj), _p(kk), E, P, _p(Ji), _p(Jj), _p(Jz), _p(xy),
                        _p(mask))
    return dict(Ji=Ji, Jj=Jj, Jz=Jz, xy=xy)


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        with open("/proc/cpuinfo") as f:
            flags = f.read()
        if " fma" not in flags or " avx2" not in flags:
            raise RuntimeError("oracle needs a host CPU with AVX2+FMA")
        _lib = ctypes.CDLL(_LIB_PATH)
        _lib.orc_ba.restype = ctypes.c_int
    return _lib


def _f(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _i(a):
    return np.ascontiguousarray(a, dtype=np.int64)


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p)


# --------------------------------------------------------------------- altcorr
def patchify_raw(net, coords, radius):
    net, coords = _f(net), _f(coords)
    n, C, H, W = net.shape
    M = coords.shape[1]
    D = 2 * radius + 2
    out = np.empty((n, M, C, D, D), np.float32)
    lib().orc_patchify_raw(_p(net), _p(coords), _p(out), n, C, H, W, M, radius)
    return out


def patchify(net, coords, radius):
    """altcorr.patchify(net, coords, radius, mode='bilinear')"""
    net, coords = _f(net), _f(coords)
    n, C, H, W = net.shape
    M = coords.shape[1]
    d = 2 * radius + 1
    out = np.empty((n, M, C, d, d), np.float32)
    lib().orc_patchify(_p(net), _p(coords), _p(out), n, C, H, W, M, radius)
    return out


def corr(fmap1, fmap2, coords, ii, jj, radius):
    """altcorr.corr: fmap1 [1,N1,C,P,P], fmap2 [1,N2,C,H,W], coords [1,E,2,P,P]
    -> [1,E,d,d,P,P] (x-offset axis first, as the reference returns it)."""
    fmap1, fmap2, coords = _f(fmap1), _f(fmap2), _f(coords)
    ii, jj = _i(ii), _i(jj)
    assert fmap1.shape[0] == 1 and fmap2.shape[0] == 1 and coords.shape[0] == 1
    _, N1, C, P, _ = fmap1.shape
    _, N2, _, H2, W2 = fmap2.shape
    E = coords.shape[1]
    assert ii.size == 0 or (ii.min() >= 0 and ii.max() < N1)
    assert jj.size == 0 or (jj.min() >= 0 and jj.max() < N2)
    d = 2 * radius + 1
    out = np.empty((1, E, d, d, P, P), np.float32)
    lib().orc_corr(_p(fmap1), _p(fmap2), _p(coords), _p(ii), _p(jj), _p(out),
                   E, C, P, H2, W2, radius)
    return out


# -------------------------------------------------------------------- lietorch
def _un(name, x, din, dout):
    x = _f(x)
    shp = x.shape[:-1]
    x2 = np.ascontiguousarray(x.reshape(-1, din))
    out = np.empty((x2.shape[0], dout), np.float32)
    getattr(lib(), name)(_p(x2), _p(out), x2.shape[0])
    return out.reshape(shp + (dout,))


def _bin(name, x, y, dx, dy, dout):
    x, y = _f(x), _f(y)
    bs = np.broadcast_shapes(x.shape[:-1], y.shape[:-1])
    x2 = np.ascontiguousarray(np.broadcast_to(x, bs + (dx,)).reshape(-1, dx))
    y2 = np.ascontiguousarray(np.broadcast_to(y, bs + (dy,)).reshape(-1, dy))
    out = np.empty((x2.shape[0], dout), np.float32)
    getattr(lib(), name)(_p(x2), _p(y2), _p(out), x2.shape[0])
    return out.reshape(bs + (dout,))


def se3_exp(a): return _un("orc_se3_exp", a, 6, 7)
def se3_log(X): return _un("orc_se3_log", X, 7, 6)
def se3_inv(X): return _un("orc_se3_inv", X, 7, 7)
def se3_mul(X, Y): return _bin("orc_se3_mul", X, Y, 7, 7, 7)
def se3_act4(X, p): return _bin("orc_se3_act4", X, p, 7, 4, 4)
def se3_adj(X, a): return _bin("orc_se3_adj", X, a, 7, 6, 6)
def se3_adjT(X, a): return _bin("orc_se3_adjT", X, a, 7, 6, 6)


# ---------------------------------------------------------------------- fastba
def transform(poses, patches, intrinsics, ii, jj, kk, tonly=False):
    """Ramp_vo.reproject == pops.transform(...).permute(0,1,4,2,3): [1,E,2,P,P]"""
    poses = _f(poses).reshape(-1, 7)
    P = patches.shape[-1]
    patches = _f(patches).reshape(-1, 3, P, P)
    intrinsics = _f(intrinsics).reshape(-1, 4)
    ii, jj, kk = _i(ii), _i(jj), _i(kk)
    E = ii.shape[0]
    out = np.empty((1, E, 2, P, P), np.float32)
    lib().orc_transform(_p(poses), _p(patches), _p(intrinsics), _p(ii), _p(jj), _p(kk),
                        _p(out), E, P, int(bool(tonly)))
    return out


def reproject(poses, patches, intrinsics, ii, jj, kk):
    """fastba.reproject: [1,E,2,P,P]"""
    poses = _f(poses).reshape(-1, 7)
    P = patches.shape[-1]
    patches = _f(patches).reshape(-1, 3, P, P)
    intrinsics = _f(intrinsics).reshape(-1, 4)
    ii, jj, kk = _i(ii), _i(jj), _i(kk)
    E = ii.shape[0]
    out = np.empty((1, E, 2, P, P), np.float32)
    lib().orc_reproject(_p(poses), _p(patches), _p(intrinsics), _p(ii), _p(jj), _p(kk),
                        _p(out), E, P)
    return out


def neighbors(ii, jj):
    ii, jj = _i(ii), _i(jj)
    E = ii.shape[0]
    ix = np.empty(E, np.int64)
    jx = np.empty(E, np.int64)
    lib().orc_neighbors(_p(ii), _p(jj), _p(ix), _p(jx), E)
    return ix, jx


def ba(poses, patches, intrinsics, target, weight, lmbda, ii, jj, kk, t0, t1, iterations=2, eff_ppf=0):
    """fastba.BA; ``poses`` and ``patches`` (float32, C-contiguous numpy) are
    updated in place like the reference.  Returns the Cholesky status.  eff_ppf > 0: the reference's
    ``eff_impl=True`` path with PPF = eff_ppf patches per frame (block-sparse E lookup, fastba/block_e.cu)."""
    for a in (poses, patches):
        assert a.dtype == np.float32 and a.flags["C_CONTIGUOUS"]
    P = patches.shape[-1]
    intrinsics = _f(intrinsics).reshape(-1, 4)
    target = _f(target).reshape(-1, 2)
    weight = _f(weight).reshape(-1, 2)
    lmbda = _f(lmbda).reshape(-1)
    ii, jj, kk = _i(ii), _i(jj), _i(kk)
    E = ii.shape[0]
    if eff_ppf:
        lib().orc_ba_eff.restype = ctypes.c_int
        return lib().orc_ba_eff(_p(poses), _p(patches), _p(intrinsics), _p(target), _p(weight),
                                _p(lmbda), _p(ii), _p(jj), _p(kk), E, P, int(t0), int(t1),
                                int(iterations), int(eff_ppf))
    return lib().orc_ba(_p(poses), _p(patches), _p(intrinsics), _p(target), _p(weight),
                        _p(lmbda), _p(ii), _p(jj), _p(kk), E, P, int(t0), int(t1),
                        int(iterations))


def segment_softmax_sum(fx, gx, group, G):
    fx, gx, group = _f(fx), _f(gx), _i(group)
    E, C = fx.shape
    y = np.empty((G, C), np.float32)
    lib().orc_segment_softmax_sum(_p(fx), _p(gx), _p(group), _p(y), E, G, C)
    return y


def event_stack(x, y, p, height, width, num_bins):
    """EventToStack_Numpy.__call__ for integer pixel coordinates (reference utils/transformers.py:128-161):
    event i goes to bin int32(float32(num_bins * i) / N) -- by INDEX, not by time stamp --, the bin images
    accumulate the polarities (+-1) and the float32 grid is cast to int8 (C cast: wraps modulo 256).
    x, y integer arrays, p int8 (+-1).  -> int8 [num_bins, height, width]"""
    x, y, p = np.asarray(x), np.asarray(y), np.asarray(p)
    n = len(x)
    grid = np.zeros((num_bins, height, width), np.float32)
    if n < 2:
        return grid.astype(np.int8)
    b = (num_bins * np.arange(n, dtype=np.float32) / n).astype(np.int32)
    xi, yi = x.astype(np.int64), y.astype(np.int64)
    m = (xi >= 0) & (yi >= 0) & (xi < width) & (yi < height)
    np.add.at(grid, (b[m], yi[m], xi[m]), p[m])
    return grid.astype(np.int32).astype(np.int8)      # wrap, as numpy's float32 -> int8 cast does on x86
