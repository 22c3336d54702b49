"""Run the reference's *unmodified* python on CPU -- build-container only.

TEST INFRASTRUCTURE.  ``/root/reference`` exists only in the build container, so
this module is used exclusively by ``oracle/make_golden.py`` (golden-vector
generation) and by CPU tests that skip when the reference is absent.  Nothing
here is copied from the reference: the reference packages are imported from
where they lie, with

  * ``sys.modules`` stubs for third-party modules the image lacks (evo,
    torchvision, h5py, hdf5plugin, yacs, numba, torch_scatter, ...), and for
    the three CUDA extensions (``cuda_corr``, ``cuda_ba``,
    ``lietorch_backends``) whose entry points are served by the C oracle
    (``oracle/ramp_oracle.c``);
  * a ``TorchFunctionMode`` that rewrites every hard-coded ``device="cuda"`` /
    ``.cuda()`` to CPU.
"""
import importlib
import os
import sys
import types

import numpy as np
import torch
from torch.overrides import TorchFunctionMode

import oracle as orc

REF_ROOT = os.environ.get("RAMP_REFERENCE", "/root/reference")


def available():
    return os.path.isdir(os.path.join(REF_ROOT, "ramp"))


# ----------------------------------------------------------------- cuda -> cpu
def _fix_dev(x):
    if isinstance(x, str) and x.startswith("cuda"):
        return "cpu"
    if isinstance(x, torch.device) and x.type == "cuda":
        return torch.device("cpu")
    return x


class CudaToCpu(TorchFunctionMode):
    def __torch_function__(self, func, types_, args=(), kwargs=None):
        kwargs = dict(kwargs or {})
        name = getattr(func, "__name__", "")
        if name == "cuda" and args and isinstance(args[0], torch.Tensor):
            return args[0]
        if "device" in kwargs:
            kwargs["device"] = _fix_dev(kwargs["device"])
        args = tuple(_fix_dev(a) for a in args)
        return func(*args, **kwargs)


# ----------------------------------------------------------------------- stubs
def _t(a):
    return torch.from_numpy(np.ascontiguousarray(a))


def _np(t):
    return t.detach().cpu().float().contiguous().numpy()


def _make_cuda_corr():
    m = types.ModuleType("cuda_corr")

    def forward(fmap1, fmap2, coords, ii, jj, radius):
        out = orc.corr(_np(fmap1), _np(fmap2), _np(coords), ii.numpy(), jj.numpy(), radius)
        return [_t(out).to(fmap1.dtype)]

    def patchify_forward(net, coords, radius):
        return [_t(orc.patchify_raw(_np(net), _np(coords), radius)).to(net.dtype)]

    m.forward = forward
    m.patchify_forward = patchify_forward
    m.backward = m.patchify_backward = None
    return m


def _make_cuda_ba():
    m = types.ModuleType("cuda_ba")

    def forward(poses, patches, intrinsics, target, weight, lmbda, ii, jj, kk, PPF, t0, t1,
                iterations, eff_impl):
        assert not eff_impl
        # in-place semantics: operate on numpy views of the caller's storage
        assert poses.is_contiguous() and patches.is_contiguous()
        p = poses.view(-1, 7).numpy()
        pt = patches.view(-1, 3, patches.shape[-2], patches.shape[-1]).numpy()
        orc.ba(p, pt, _np(intrinsics), _np(target), _np(weight), _np(lmbda), ii.numpy(),
               jj.numpy(), kk.numpy(), t0, t1, iterations)
        return []

    def neighbors(ii, jj):
        ix, jx = orc.neighbors(ii.numpy(), jj.numpy())
        return [_t(ix), _t(jx)]

    def reproject(poses, patches, intrinsics, ii, jj, kk):
        return _t(orc.reproject(_np(poses), _np(patches), _np(intrinsics), ii.numpy(),
                                jj.numpy(), kk.numpy()))

    m.forward, m.neighbors, m.reproject = forward, neighbors, reproject
    m.solve_system = None
    return m


def _make_lietorch_backends():
    m = types.ModuleType("lietorch_backends")

    def _chk(g):
        assert g == 3, "oracle restates SE3 (group_id 3) only"

    # float64 tensors (the fp64 pin of the bundle-adjustment algebra, oracle/make_golden.py::gen_ba_f64_pin) are served by
    # the fp64 build of the same oracle source, everything else by the fp32 restatement
    def _d(*ts):
        return all(t.dtype == torch.float64 for t in ts)

    def _n(t):
        return t.detach().cpu().contiguous().numpy() if t.dtype == torch.float64 else _np(t)

    def expm(g, a): _chk(g); return _t((orc.se3_exp_f64 if _d(a) else orc.se3_exp)(_n(a)))
    def logm(g, X): _chk(g); return _t((orc.se3_log_f64 if _d(X) else orc.se3_log)(_n(X)))
    def inv(g, X): _chk(g); return _t((orc.se3_inv_f64 if _d(X) else orc.se3_inv)(_n(X)))
    def mul(g, X, Y): _chk(g); return _t((orc.se3_mul_f64 if _d(X, Y) else orc.se3_mul)(_n(X), _n(Y)))
    def adj(g, X, a): _chk(g); return _t((orc.se3_adj_f64 if _d(X, a) else orc.se3_adj)(_n(X), _n(a)))
    def adjT(g, X, a): _chk(g); return _t((orc.se3_adjT_f64 if _d(X, a) else orc.se3_adjT)(_n(X), _n(a)))
    def act4(g, X, p): _chk(g); return _t((orc.se3_act4_f64 if _d(X, p) else orc.se3_act4)(_n(X), _n(p)))

    def act(g, X, p):
        _chk(g)
        f64 = _d(X, p)
        p4 = np.concatenate([_n(p), np.ones(p.shape[:-1] + (1,), np.float64 if f64 else np.float32)], -1)
        return _t((orc.se3_act4_f64 if f64 else orc.se3_act4)(_n(X), p4)[..., :3])

    for f in (expm, logm, inv, mul, adj, adjT, act, act4):
        setattr(m, f.__name__, f)
    for name in ("expm_backward", "logm_backward", "inv_backward", "mul_backward",
                 "adj_backward", "adjT_backward", "act_backward", "act4_backward",
                 "as_matrix", "projector", "Jinv"):
        setattr(m, name, None)
    return m


def _make_torch_scatter():
    m = types.ModuleType("torch_scatter")

    def scatter_sum(src, index, dim=-1, out=None, dim_size=None):
        dim = dim % src.dim()
        if dim_size is None:
            dim_size = int(index.max()) + 1 if index.numel() else 0
        shape = list(src.shape)
        shape[dim] = dim_size
        res = torch.zeros(shape, dtype=src.dtype)
        return res.index_add_(dim, index, src)

    def scatter_softmax(src, index, dim=-1):
        dim = dim % src.dim()
        n = int(index.max()) + 1
        shape = list(src.shape)
        shape[dim] = n
        idx = index.view([-1 if d == dim else 1 for d in range(src.dim())]).expand_as(src)
        mx = torch.full(shape, float("-inf"), dtype=src.dtype).scatter_reduce_(
            dim, idx, src, "amax", include_self=True)
        ex = (src - mx.index_select(dim, index)).exp()
        sm = scatter_sum(ex, index, dim, dim_size=n)
        return ex / sm.index_select(dim, index)

    m.scatter_sum, m.scatter_softmax = scatter_sum, scatter_softmax
    return m


class _AttrDict(dict):
    """stand-in for yacs.config.CfgNode: attribute access + merge_from_file"""
    __getattr__ = dict.__getitem__
    __setattr__ = dict.__setitem__

    def merge_from_file(self, path):
        import yaml
        with open(path) as f:
            self.update(yaml.safe_load(f))

    def clone(self):
        return _AttrDict(self)


def _dummy_module(name, **attrs):
    m = types.ModuleType(name)
    m.__path__ = []
    for k, v in attrs.items():
        setattr(m, k, v)
    return m


_loaded = None


def load():
    """import the reference packages (idempotent); returns a namespace."""
    global _loaded
    if _loaded is not None:
        return _loaded
    if not available():
        raise RuntimeError("reference checkout not present at %s" % REF_ROOT)
    sys.dont_write_bytecode = True
    if REF_ROOT not in sys.path:
        sys.path.insert(0, REF_ROOT)

    class _Any:
        def __init__(self, *a, **k): pass
        def __call__(self, *a, **k): return self
        def __getattr__(self, k): return _Any()

    stubs = {
        "evo": _dummy_module("evo"), "evo.core": _dummy_module("evo.core"),
        "evo.core.trajectory": _dummy_module("evo.core.trajectory", PoseTrajectory3D=object),
        "torchvision": _dummy_module("torchvision"),
        "torchvision.transforms": _dummy_module("torchvision.transforms", Grayscale=object),
        "h5py": _dummy_module("h5py", File=_Any), "hdf5plugin": _dummy_module("hdf5plugin"),
        "yacs": _dummy_module("yacs"),
        "yacs.config": _dummy_module("yacs.config", CfgNode=_AttrDict),
        "numba": _dummy_module("numba", jit=lambda *a, **k: (lambda f: f)),
        "torch_scatter": _make_torch_scatter(),
        "cuda_corr": _make_cuda_corr(), "cuda_ba": _make_cuda_ba(),
        "lietorch_backends": _make_lietorch_backends(),
    }
    for opt in ("PIL", "PIL.Image", "matplotlib", "matplotlib.pyplot", "pandas", "tqdm",
                "sklearn", "sklearn.gaussian_process", "sklearn.gaussian_process.kernels"):
        try:
            importlib.import_module(opt)
        except Exception:
            stubs[opt] = _dummy_module(opt, **{k: _Any for k in (
                "Image", "Matern", "WhiteKernel", "ConstantKernel", "gaussian_process", "tqdm")})
    for k, v in stubs.items():
        sys.modules.setdefault(k, v)

    ns = types.SimpleNamespace()
    with CudaToCpu():
        ns.extractor = importlib.import_module("ramp.extractor")
        ns.pops = importlib.import_module("ramp.projective_ops")
        ns.lietorch = importlib.import_module("ramp.lietorch")
        ns.altcorr = importlib.import_module("ramp.altcorr.correlation")
        ns.fastba = importlib.import_module("ramp.fastba.ba")
        ns.blocks = importlib.import_module("ramp.blocks")
        ns.utils = importlib.import_module("ramp.utils")
        ns.net = importlib.import_module("ramp.net")
        ns.ba = importlib.import_module("ramp.ba")
        ns.Ramp_vo = importlib.import_module("ramp.Ramp_vo")
        ns.config = importlib.import_module("ramp.config")
    ns.CfgNode = _AttrDict
    ns.mode = CudaToCpu
    _loaded = ns
    return ns
