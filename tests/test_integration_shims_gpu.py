"""INTEGRATION.md section B is code under test: the three ctypes shim modules a maintainer of the reference would drop
in place of its pybind extensions (`cuda_corr`, `cuda_ba`, `lietorch_backends`; reference signatures:
ramp/altcorr/correlation.cpp:28-63, ramp/fastba/ba.cpp:32-59, ramp/lietorch/src/lietorch.cpp:286-316) are cut out of
the document, imported as modules of those names and driven with the reference's tensor shapes against the CPU oracle.
"""
import os
import re
import sys
import types

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SHIM_NAMES = ("cuda_corr", "cuda_ba", "lietorch_backends")


def shim_sources():
    """the python blocks of section B, keyed by the module name in their first comment line"""
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    sec = text[text.index("## B. "):text.index("## Ownership")]
    blocks = re.findall(r"```python\n(.*?)```", sec, re.S)
    out = {}
    for b in blocks:
        m = re.match(r"# (\w+)\.py", b)
        assert m, "every block of section B starts with '# <module>.py'"
        out[m.group(1)] = b
    assert tuple(out) == SHIM_NAMES, tuple(out)
    return out


def test_shim_blocks_only_use_exported_entry_points():
    """(CPU) every L.ramp_* name the documented shims call is declared in include/ramp_hip.h and exported by the .so"""
    from rampvo_amd import _lib
    lib = _lib.lib()
    header = open(os.path.join(ROOT, "include", "ramp_hip.h")).read()
    used = set()
    for src in shim_sources().values():
        used |= set(re.findall(r"L\.(ramp_\w+)", src)) | set(re.findall(r'"(ramp_se3_\w+)"', src))
    assert len(used) >= 13, used
    for name in sorted(used):
        assert re.search(r"\b%s\s*\(" % name, header), name + " is not declared in include/ramp_hip.h"
        assert hasattr(lib, name), name + " is not exported"


@pytest.fixture(scope="module")
def shims():
    from rampvo_amd import _lib
    os.environ["RAMP_HIP_LIB"] = _lib.LIB_PATH
    saved = {n: sys.modules.get(n) for n in SHIM_NAMES}
    mods = {}
    for name, src in shim_sources().items():           # in document order: cuda_ba / lietorch import from cuda_corr
        mod = types.ModuleType(name)
        mod.__file__ = "INTEGRATION.md::" + name
        sys.modules[name] = mod
        exec(compile(src, mod.__file__, "exec"), mod.__dict__)
        mods[name] = mod
    yield mods
    for n, m in saved.items():
        if m is None:
            sys.modules.pop(n, None)
        else:
            sys.modules[n] = m


def cu(a):
    import torch
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


@pytest.mark.gpu
def test_cuda_corr_shim_forward_and_patchify(shims):
    """cuda_corr.forward -> [1,E,7,7,3,3] blended + permuted (the reference blends in Python, altcorr/correlation.py:
    6-13 over correlation_kernel.cu:193-233); cuda_corr.patchify_forward -> raw [n,M,C,D,D] windows"""
    import oracle as orc
    from scenes import corr_case
    cc = shims["cuda_corr"]
    fmap1, fmap2, coords, ii, jj = corr_case(seed=21, E=64, N1=96 * 2, N2=8, H=30, W=40)
    out, = cc.forward(cu(fmap1), cu(fmap2), cu(coords), cu(ii), cu(jj), 3)
    assert tuple(out.shape) == (1, 64, 7, 7, 3, 3)
    ref = orc.corr(fmap1, fmap2, coords, ii, jj, 3)[0]
    got = out[0].cpu().numpy()
    assert np.array_equal(np.isnan(got), np.isnan(ref))
    assert np.nanmax(np.abs(got - ref)) <= 1e-5
    rng = np.random.default_rng(5)
    for C, R, (H, W) in ((128, 1, (30, 40)), (384, 0, (30, 40)), (3, 0, (120, 160))):
        net = rng.normal(size=(1, C, H, W)).astype(np.float32)
        xy = np.stack([rng.uniform(-2, W + 2, (1, 96)), rng.uniform(-2, H + 2, (1, 96))], -1).astype(np.float32)
        raw, = cc.patchify_forward(cu(net), cu(xy), R)
        assert tuple(raw.shape) == (1, 96, C, 2 * R + 2, 2 * R + 2)
        assert np.array_equal(raw.cpu().numpy(), orc.patchify_raw(net, xy, R))        # a gather: bit exact


@pytest.mark.gpu
def test_cuda_ba_shim_forward_neighbors_reproject(shims):
    import torch
    import oracle as orc
    from scenes import ba_scene
    ba = shims["cuda_ba"]
    s = ba_scene(seed=11, n_frames=10, M=12, lifetime=4, n_total_frames=16)
    p_ref, pt_ref = s["poses"].copy(), s["patches"].copy()
    orc.ba(p_ref, pt_ref, s["intr"], s["target"], s["weight"], s["lmbda"], s["ii"], s["jj"], s["kk"], 2, 10, 2)
    # the reference's shapes: poses [1,N,7], patches [1,N*M,3,3,3], intrinsics [1,N,4], target / weight [1,E,2]
    poses, patches = cu(s["poses"])[None], cu(s["patches"])[None]
    r = ba.forward(poses, patches, cu(s["intr"])[None], cu(s["target"])[None], cu(s["weight"])[None], cu(s["lmbda"]),
                   cu(s["ii"]), cu(s["jj"]), cu(s["kk"]), 12, 2, 10, 2, False)
    assert r == []
    torch.cuda.synchronize()
    step = np.abs(p_ref - s["poses"]).max()
    assert np.abs(poses[0].cpu().numpy() - p_ref).max() <= 1e-4 * max(step, 1.0)
    d_ref, d = pt_ref[:, 2, 1, 1], patches[0, :, 2, 1, 1].cpu().numpy()
    assert np.abs(d - d_ref).max() <= 1e-4 * np.abs(d_ref).max()
    # neighbors(kk, jj) as ramp/net.py:77 calls it: int64 outputs, exact
    ix, jx = ba.neighbors(cu(s["kk"]), cu(s["jj"]))
    rix, rjx = orc.neighbors(s["kk"], s["jj"])
    assert ix.dtype == torch.int64 and np.array_equal(ix.cpu().numpy(), rix) and np.array_equal(jx.cpu().numpy(), rjx)
    out = ba.reproject(cu(s["poses"])[None], cu(s["patches"])[None], cu(s["intr"])[None], cu(s["ii"]), cu(s["jj"]),
                       cu(s["kk"]))
    ref = orc.reproject(s["poses"], s["patches"], s["intr"], s["ii"], s["jj"], s["kk"])
    assert tuple(out.shape) == (1, len(s["ii"]), 2, 3, 3)
    got = out.cpu().numpy()
    fin = np.isfinite(ref) & (np.abs(ref) < 1e6)
    assert np.abs(got[fin] - ref[fin]).max() / np.abs(ref[fin]).max() < 1e-5


@pytest.mark.gpu
def test_lietorch_backends_shim(shims):
    import oracle as orc
    lb = shims["lietorch_backends"]
    rng = np.random.default_rng(2)
    a = (0.4 * rng.normal(size=(300, 6))).astype(np.float32)
    a[0] = 0
    a[1, 3:] = 1e-7
    X = orc.se3_exp(a)
    Y = orc.se3_exp((0.3 * rng.normal(size=(300, 6))).astype(np.float32))
    p = rng.normal(size=(300, 4)).astype(np.float32)
    b = rng.normal(size=(300, 6)).astype(np.float32)
    g = 3                                              # lietorch's group id of SE3
    assert np.abs(lb.expm(g, cu(a)).cpu().numpy() - X).max() < 2e-6
    assert np.abs(lb.logm(g, cu(X)).cpu().numpy() - orc.se3_log(X)).max() < 5e-6
    assert np.abs(lb.inv(g, cu(X)).cpu().numpy() - orc.se3_inv(X)).max() < 2e-6
    assert np.abs(lb.mul(g, cu(X), cu(Y)).cpu().numpy() - orc.se3_mul(X, Y)).max() < 2e-6
    assert np.abs(lb.act4(g, cu(X), cu(p)).cpu().numpy() - orc.se3_act4(X, p)).max() < 5e-6
    assert np.abs(lb.adjT(g, cu(X), cu(b)).cpu().numpy() - orc.se3_adjT(X, b)).max() < 1e-5
