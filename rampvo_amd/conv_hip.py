"""HIP back-end of the RAMP encoder: fused per-pixel LSTM / super-state kernel and the
implicit-GEMM MFMA conv towers (csrc/conv.hip).  Activations are NHWC float32 tensors
[H, W, C]; InstanceNorm + ReLU are never materialised on their own -- they ride on the
consumer conv's load path (``pre``) or on the residual-block tail kernel.
"""
import ctypes
import threading
import os

import torch
import torch.nn as nn

from . import _lib
from ._lib import RAMP_F32, check, lib, ptr, stream

def _cache(module):
    """packed-weight cache living ON the module (a process-wide dict keyed by id() would hand a new
    module the packs of a dead one whose id and storage addresses got recycled)"""
    c = module.__dict__.get("_ramp_pack")
    if c is None:
        c = {}
        object.__setattr__(module, "_ramp_pack", c)
    return c


def available():
    try:
        return hasattr(lib(), "ramp_conv2d_nhwc")
    except Exception:
        return False


# ------------------------------------------------------------------ weight packing
def pack_conv_weight(conv, mode="f32"):
    """[Cout,Cin,KH,KW] -> MFMA fragment order [tap][Cin/KC][Cout/16][64 lanes][CPL]:
    lane (q = lane>>4, j = lane&15) holds W[16*nt + j][KC*ch + CPL*q + s], s < CPL.
    mode "f32": KC=16, CPL=4 fp32;  "f16": KC=32, CPL=8 half;  "f16_first": KC=16, CPL=4 half;
    "f8": the "f16" order with OCP e4m3 bytes of w * w_scale, w_scale = 448 / max|w| (-> third return value)"""
    w = conv.weight
    key = (mode, w._version, w.device, w.data_ptr())
    hit = _cache(conv).get(mode)
    if hit is not None and hit[0] == key:
        return hit[1:] if mode == "f8" else (hit[1], hit[2])
    cout, cin, kh, kw = w.shape
    kc, cpl = (32, 8) if (mode in ("f16", "f8") or (mode == "x3" and cin >= 32)) else (16, 4)
    cin_p = (cin + kc - 1) // kc * kc
    wp = torch.zeros(cout, cin_p, kh, kw, dtype=torch.float32, device=w.device)
    wp[:, :cin] = w.detach().float()
    t = wp.permute(2, 3, 1, 0).reshape(kh * kw, cin_p // kc, 4, cpl, cout // 16, 16)   # tap, ch, q, s, nt, j
    t = t.permute(0, 1, 4, 2, 5, 3).contiguous()                                        # tap, ch, nt, q, j, s
    bias = conv.bias.detach().float().contiguous() if conv.bias is not None else None
    if mode == "f8":
        w_scale = 448.0 / max(float(w.detach().abs().max()), 1e-30)
        t = (t * w_scale).clamp(-448.0, 448.0).to(torch.float8_e4m3fn).view(torch.uint8)
        _cache(conv)[mode] = (key, t, bias, w_scale)
        return t, bias, w_scale
    if mode == "x3":
        # exact-class products on the f16 matrix cores (csrc/conv.hip::conv_x3_kernel): fragments of the THREE fp16 parts of
        # W 2^s -- w0 = fp16(W 2^s), w1 = fp16((W 2^s - w0) 2^11), w2 = fp16((W 2^s - w0 - w1 2^-11) 2^22), s the power of two
        # that puts max |W| 2^s into [2^12, 2^13) -- followed by the exact inverse 2^-s as one float
        import math
        amax = float(w.detach().abs().max())
        e = (12 - math.floor(math.log2(amax))) if amax > 0 else 0
        ws = t.double() * (2.0 ** e)
        w0 = ws.half()
        r1 = ws - w0.double()
        w1 = (r1 * 2048.0).half()
        r2 = r1 - w1.double() / 2048.0
        w2 = (r2 * 4194304.0).half()
        tail = torch.tensor([2.0 ** -e], dtype=torch.float32, device=w.device).view(torch.float16)
        t = torch.cat([w0.flatten(), w1.flatten(), w2.flatten(), tail]).contiguous()
        _cache(conv)[mode] = (key, t, bias)
        return t, bias
    if mode != "f32":
        t = t.half()
    _cache(conv)[mode] = (key, t, bias)
    return t, bias


def pack_lstm_mfma(enc):
    """per-lane A fragments / accumulator-init biases of lstm_superstate_mfma_kernel (layout LM_* in
    csrc/conv.hip): [fragment][64 lanes] float32.  MFMA 16x16x4: lane l supplies A[row l&15][k l>>4]."""
    key = tuple((p.data_ptr(), p._version) for p in enc.events_convlstm.parameters()) + \
        tuple((p.data_ptr(), p._version) for p in enc.image_convlstm.parameters()) + \
        tuple((p.data_ptr(), p._version) for p in enc.superstate_encoder.parameters())
    hit = _cache(enc).get("lstm_mfma")
    if hit is not None and hit[0] == key:
        return hit[1]
    dev = enc.superstate_encoder.weight.device
    lane = torch.arange(64)
    i, q = lane & 15, lane >> 4
    frags = []

    def lstm_frags(lstm, cin, ksteps):
        w_ih, w_hh = lstm.weight_ih_l0.detach().float().cpu(), lstm.weight_hh_l0.detach().float().cpu()
        b = (lstm.bias_ih_l0 + lstm.bias_hh_l0).detach().float().cpu()
        # full K matrix [60 rows][16 (h, unit 15 = 0) + 8 (x, zero padded)]
        wk = torch.zeros(60, 24)
        wk[:, :15] = w_hh
        wk[:, 16:16 + cin] = w_ih
        out, bias = [], []
        for t in range(4):
            unit, gate = 4 * t + (i >> 2), i & 3
            row = gate * 15 + unit.clamp(max=14)
            ok = (unit < 15).float()
            for s4 in range(ksteps):
                out.append(wk[row, 4 * s4 + q] * ok)
        for t in range(4):
            for r in range(4):                     # accumulator register r of lane (q, j): row 4q+r of tile t
                unit = 4 * t + q
                bias.append(b[r * 15 + unit.clamp(max=14)] * (unit < 15).float())
        return out, bias

    ev_f, ev_b = lstm_frags(enc.events_convlstm, 5, 6)
    im_f, im_b = lstm_frags(enc.image_convlstm, 3, 5)
    wss = torch.zeros(16, 32)
    w = enc.superstate_encoder.weight.detach().float().cpu().view(15, 30)
    wss[:15, :15] = w[:, :15]          # columns 0..15: previous super-state channels
    wss[:15, 16:31] = w[:, 15:]        # columns 16..31: embedding units
    ss_f = [wss[i, 4 * q + st] for st in range(4)] + [wss[i, 16 + 4 * t + q] for t in range(4)]
    bs = torch.zeros(16)
    bs[:15] = enc.superstate_encoder.bias.detach().float().cpu()
    ss_b = [bs[4 * q + r] for r in range(4)]
    frags = ev_f + im_f + ss_f + ev_b + im_b + ss_b
    assert len(frags) == 24 + 20 + 8 + 16 + 16 + 4
    wf = torch.stack(frags).contiguous().to(dev)
    _cache(enc)["lstm_mfma"] = (key, wf)
    return wf


# ------------------------------------------------------------------------ primitives
IN_ACC_R = 8             # include/ramp_hip.h::RAMP_IN_ACC_R
_IN_ACC = True           # (False: per-layer ramp_in_stats_finalize launches -- round 3's A/B)


class Pending:
    """a raw conv output whose InstanceNorm has not been applied yet; ``relu``: the norm is followed by a ReLU
    (conv1 / the residual block's convs) or not (the downsample path's norm3).  The statistics are either finalised
    (scale, shift) arrays or -- accumulator mode -- the replicated fixed-point sums ``acc`` [R][C][2] int64 with the pixel
    count and eps, which the consuming kernels reduce themselves; asking such a Pending for ``scale`` / ``shift``
    finalises it in a launch of its own (consumers without the accumulator path)."""
    __slots__ = ("raw", "_scale", "_shift", "relu", "acc", "count", "eps")

    def __init__(self, raw, scale, shift, relu=True, acc=None, count=0.0, eps=1e-5):
        self.raw, self._scale, self._shift, self.relu = raw, scale, shift, relu
        self.acc, self.count, self.eps = acc, float(count), float(eps)

    def _finalise(self):
        if self._scale is None:
            C = self.raw.shape[-1]
            ws = torch.empty(2, C, dtype=torch.float32, device=self.raw.device)
            check(lib().ramp_in_acc_finalize(ptr(self.acc), C, self.count, self.eps, ptr(ws[0]), ptr(ws[1]), stream()),
                  "ramp_in_acc_finalize")
            self._scale, self._shift = ws[0], ws[1]

    @property
    def scale(self):
        self._finalise()
        return self._scale

    @property
    def shift(self):
        self._finalise()
        return self._shift


class _AccArena:
    """the accumulators of one tower pass: ONE zeroed int64 allocation (one memset for all layers), slots of
    [R][128][2] handed out in order"""
    SLOTS = 16

    def __init__(self, device, zeroed=True):
        # zeroed=False: the caller zeroes buf itself before the pass (lstm_superstate_step: inside the front end's first launch)
        make = torch.zeros if zeroed else torch.empty
        self.buf = make(self.SLOTS, IN_ACC_R * 128 * 2, dtype=torch.int64, device=device)
        self.used = 0

    def slot(self, C):
        if self.used >= self.SLOTS or C > 128:
            return None
        v = self.buf[self.used][:IN_ACC_R * C * 2]
        self.used += 1
        return v


class _ArenaScope(threading.local):
    """the accumulator arena of the tower pass running in THIS thread (None outside one): two trackers on two threads do
    not see each other's"""
    cur = None
    prepared = None        # an arena the caller has already zeroed for the next basic_encoder4_towers pass of this thread


_scope = _ArenaScope()


def set_arena(a):
    """(tests) make `a` the arena the next conv2d_towers calls of this thread take their accumulators from"""
    _scope.cur = a


class ConvJob(ctypes.Structure):
    """include/ramp_hip.h::ramp_conv_job"""
    _fields_ = [("x", ctypes.c_void_p), ("wpk", ctypes.c_void_p), ("bias", ctypes.c_void_p),
                ("pre_scale", ctypes.c_void_p), ("pre_shift", ctypes.c_void_p), ("res", ctypes.c_void_p),
                ("y", ctypes.c_void_p), ("stats", ctypes.c_void_p),
                ("Cout", ctypes.c_int32), ("relu", ctypes.c_int32), ("out_scale", ctypes.c_float),
                ("act_scale", ctypes.c_float), ("w_scale", ctypes.c_float),
                ("acc_out", ctypes.c_void_p), ("acc_in", ctypes.c_void_p), ("in_count", ctypes.c_float),
                ("in_eps", ctypes.c_float), ("x2", ctypes.c_void_p), ("c0", ctypes.c_int32),
                ("skip", ctypes.c_void_p), ("acc_skip", ctypes.c_void_p), ("skip_count", ctypes.c_float),
                ("skip_eps", ctypes.c_float), ("skip_relu", ctypes.c_int32), ("mat", ctypes.c_void_p)]


_TAIL_FUSE = True        # (False: a norm_add_relu launch per residual block -- round 4's A/B)


class Tail:
    """a residual block's output relu(skip' + relu(norm(y))) that has not been computed yet (fp16 towers in accumulator
    mode): the LDS-tiled conv kernel forms it while it loads its input tile (ramp_conv_job.skip), and a stride-1 consumer
    writes it out on the way (``mat``) for the next block's skip.  ``tensor()`` computes it with the launch of its own
    (consumers without that path)."""
    __slots__ = ("y", "skip", "out", "written")

    def __init__(self, y, skip):
        self.y, self.skip, self.out, self.written = y, skip, None, False

    def tensor(self):
        if not self.written:
            self.out = norm_add_relu(self.y, self.skip, fuse=False)
            self.written = True
        return self.out

    def fusable(self):
        sp = isinstance(self.skip, Pending)
        return (self.y.raw.dtype == torch.float16 and self.y.acc is not None and self.y._scale is None and self.y.relu
                and self.y.raw.shape[-1] <= 128 and (not sp or (self.skip.acc is not None and self.skip._scale is None)))


class Pair:
    """a channel concatenation that is never materialised: (a [H,W,C0], b [H,W,C1]) read as [H,W,C0+C1] by the LDS-tiled
    conv kernel (ramp_conv_job.x2)"""
    __slots__ = ("a", "b")

    def __init__(self, a, b):
        assert a.shape[:2] == b.shape[:2] and a.dtype == b.dtype and a.is_contiguous() and b.is_contiguous()
        self.a, self.b = a, b

    def cat(self):
        return torch.cat((self.a, self.b), dim=-1)


# fp8 variant: activations are multiplied by this before the e4m3 conversion (saturating at 448 / FP8_ACT_SCALE = 56;
# post-InstanceNorm / ReLU activations and the random-init plain tower stay well inside; e4m3's relative precision,
# 2^-4, does not depend on the scale)
FP8_ACT_SCALE = 8.0
# fp32 towers: 3x3 / 7x7 layers at fp32 accuracy on the f16 matrix cores (conv_x3_kernel) instead of the f32 MFMA's direct
# kernel; False: the exact-product f32 MFMA everywhere (tests compare the two)
X3 = os.environ.get("RAMP_CONV_X3", "1") != "0"      # env: 0 = the fp32 towers on the f32 MFMA (exact products)


def _conv_mode(x, half, direct=False):
    if not half:
        return "f32", RAMP_F32, torch.float32
    if x.dtype == torch.float32:
        mode, code = "f16_first", _lib.RAMP_F16 | _lib.RAMP_IN_F32
    else:
        mode, code = "f16", _lib.RAMP_F16
    if direct:
        code |= _lib.RAMP_CONV_DIRECT        # the one-round-trip-per-tap kernel (A/B test of the LDS-tiled one)
    return mode, code, torch.float16


def conv2d(x, conv, pre=None, res=None, relu=False, want_stats=False, out_scale=1.0, eps=1e-5, half=False,
           direct=False):
    """x [H,W,Cin] NHWC (or a Pending: normalise+ReLU on load).  fp32 in/out, or with ``half``:
    half out and half in (fp32 in allowed for the 16-channel first layer).  Returns y [OH,OW,Cout],
    or Pending(y, scale, shift) when want_stats (InstanceNorm statistics of y, always fp32)."""
    if isinstance(x, Tail):
        x = x.tensor()
    if isinstance(x, Pending):
        assert x.relu
        pre, x = (x.scale, x.shift), x.raw
    H, W, Cin = x.shape
    mode, code, odt = _conv_mode(x, half, direct)
    cout, _, kh, kw = conv.weight.shape
    stride = conv.stride[0]
    if mode == "f32" and X3 and kh == kw and lib().ramp_conv2d_stats_blocks(H, W, Cin, cout, kh, stride,
                                                                           RAMP_F32 | _lib.RAMP_CONV_X3) > 0:
        mode, code = "x3", RAMP_F32 | _lib.RAMP_CONV_X3      # the layer shapes conv_x3_kernel covers (the towers' 3x3 / 7x7 layers)
    wpk, bias = pack_conv_weight(conv, mode)
    kc = 32 if mode == "f16" else 16
    assert (mode == "x3" and Cin == (conv.weight.shape[1] + 15) // 16 * 16) or (mode != "x3" and Cin == wpk.shape[1] * kc)
    assert conv.padding[0] == kh // 2 and x.is_contiguous()
    assert res is None or (res.is_contiguous() and res.dtype == odt)
    OH = (H + 2 * (kh // 2) - kh) // stride + 1
    OW = (W + 2 * (kw // 2) - kw) // stride + 1
    y = torch.empty(OH, OW, cout, dtype=odt, device=x.device)
    nblk = lib().ramp_conv2d_stats_blocks(H, W, Cin, cout, kh, stride, code)
    assert nblk > 0
    stats = torch.empty(cout, 2, nblk, dtype=torch.float32, device=x.device) if want_stats else None
    check(lib().ramp_conv2d_nhwc(ptr(x), ptr(wpk), ptr(bias), ptr(pre[0]) if pre else None,
                                 ptr(pre[1]) if pre else None, ptr(res), ptr(y), ptr(stats), H, W, Cin, cout,
                                 kh, kw, stride, int(relu), float(out_scale), code, stream()),
          "ramp_conv2d_nhwc")
    if not want_stats:
        return y
    scale = torch.empty(cout, dtype=torch.float32, device=x.device)
    shift = torch.empty(cout, dtype=torch.float32, device=x.device)
    check(lib().ramp_in_stats_finalize(ptr(stats), nblk, cout, float(OH * OW), float(eps), ptr(scale), ptr(shift),
                                       stream()), "ramp_in_stats_finalize")
    return Pending(y, scale, shift)


def conv2d_towers(jobs, half, fp8=False):
    """one layer of every tower: ``jobs`` = [dict(x=, conv=, res=None, relu=False, want_stats=False, out_scale=1.0,
    eps=1e-5)] with the same layer shape.  Two fp16 towers go out as ONE launch of the LDS-tiled kernel
    (ramp_conv2d_nhwc_multi) followed by the statistics' finalize launch where a tower has a norm; anything else
    (fp32 towers, shapes the tiled kernel does not cover) is the per-tower conv2d() above.  Returns one tensor /
    Pending per job.  (Finalising the statistics inside the conv launch -- last workgroup per tile row, then last
    row -- was measured and dropped: the device-scope release every workgroup needs before its ticket writes the
    XCD's L2 back, 1.24 ms per front end instead of 0.47.)"""
    def single():
        for j in jobs:
            if isinstance(j.get("res"), Tail):
                j["res"] = j["res"].tensor()
        return [conv2d(j["x"].cat() if isinstance(j["x"], Pair) else j["x"], j["conv"], res=j.get("res"),
                       relu=j.get("relu", False), want_stats=j.get("want_stats", False),
                       out_scale=j.get("out_scale", 1.0), eps=j.get("eps", 1e-5), half=half) for j in jobs]
    if not half or len(jobs) > 2:
        return single()
    x0 = jobs[0]["x"]
    x0 = x0.y.raw if isinstance(x0, Tail) else x0.raw if isinstance(x0, Pending) else x0
    paired = isinstance(x0, Pair)
    if paired:
        if not all(isinstance(j["x"], Pair) and j["x"].a.shape == x0.a.shape and j["x"].b.shape == x0.b.shape for j in jobs):
            return single()
        H, W = x0.a.shape[:2]
        Cin = x0.a.shape[2] + x0.b.shape[2]
        x0 = x0.a
    else:
        H, W, Cin = x0.shape
    mode, code, odt = _conv_mode(x0, True)
    use8 = bool(fp8) and mode == "f16"            # (the fp32-input first layer stays on the f16 MFMA)
    if use8:
        code |= _lib.RAMP_CONV_FP8
    c0 = jobs[0]["conv"]
    kh, stride = c0.weight.shape[2], c0.stride[0]
    OH = (H + 2 * (kh // 2) - kh) // stride + 1
    OW = (W + 2 * (kh // 2) - kh) // stride + 1
    nblk = lib().ramp_conv2d_stats_blocks(H, W, Cin, c0.weight.shape[0], kh, stride, code)
    tiles_y = (OH + 7) // 8
    if nblk != tiles_y * ((OW + 15) // 16):
        return single()                          # not a tiled-kernel layer shape
    arr = (ConvJob * len(jobs))()
    outs, finalize, written = [], [], []
    for t, j in enumerate(jobs):
        x, conv = j["x"], j["conv"]
        pre = acc_in = None
        x2 = None
        tail = None
        if paired:
            x, x2 = x.a, x.b
        if isinstance(x, Tail):
            if x.written or not x.fusable() or use8 or _scope.cur is None:   # (fp8 / per-block statistics: no fused instance)
                x = x.tensor()
            else:
                tail, x = x, x.y
        if isinstance(x, Pending):
            assert x.relu
            if x.acc is not None and x._scale is None and Cin <= 128:
                acc_in = x
            else:
                pre = (x.scale, x.shift)
            x = x.raw
        w_scale = 0.0
        if use8:
            wpk, bias, w_scale = pack_conv_weight(conv, "f8")
        else:
            wpk, bias = pack_conv_weight(conv, mode)
        cout = conv.weight.shape[0]
        assert tuple(x.shape) == (H, W, Cin - (x2.shape[2] if x2 is not None else 0)) and x.dtype == x0.dtype and x.is_contiguous()
        assert conv.weight.shape[2] == kh and conv.stride[0] == stride and conv.padding[0] == kh // 2
        assert Cin == wpk.shape[1] * (32 if mode == "f16" else 16)
        res = j.get("res")
        if isinstance(res, Tail):
            res = res.tensor()
        assert res is None or (res.is_contiguous() and res.dtype == odt)
        y = torch.empty(OH, OW, cout, dtype=odt, device=x.device)
        a = arr[t]
        a.x, a.wpk, a.bias = ptr(x), ptr(wpk), ptr(bias)
        a.x2, a.c0 = (ptr(x2), x.shape[2]) if x2 is not None else (None, 0)
        a.pre_scale, a.pre_shift = (ptr(pre[0]), ptr(pre[1])) if pre else (None, None)
        a.acc_in, a.in_count, a.in_eps = (ptr(acc_in.acc), acc_in.count, acc_in.eps) if acc_in else (None, 0.0, 0.0)
        a.skip = a.acc_skip = a.mat = None
        a.skip_count = a.skip_eps = 0.0
        a.skip_relu = 0
        if tail is not None:
            assert acc_in is not None
            sk = tail.skip
            if isinstance(sk, Tail):
                sk = sk.tensor()
            if isinstance(sk, Pending):
                a.skip, a.acc_skip, a.skip_count, a.skip_eps, a.skip_relu = ptr(sk.raw), ptr(sk.acc), sk.count, sk.eps, int(sk.relu)
            else:
                assert sk.is_contiguous() and sk.dtype == x.dtype and sk.shape == x.shape
                a.skip = ptr(sk)
            if stride == 1 and j.get("keep", True):
                tail.out = torch.empty_like(x)
                a.mat = ptr(tail.out)
                written.append(tail)
        a.acc_out = None
        a.res, a.y = ptr(res), ptr(y)
        a.Cout, a.relu, a.out_scale = cout, int(j.get("relu", False)), float(j.get("out_scale", 1.0))
        a.act_scale, a.w_scale = (FP8_ACT_SCALE, w_scale) if use8 else (0.0, 0.0)
        acc = _scope.cur.slot(cout) if (j.get("want_stats", False) and _scope.cur is not None) else None
        if acc is not None:
            a.stats, a.acc_out = None, ptr(acc)
            outs.append(Pending(y, None, None, acc=acc, count=OH * OW, eps=j.get("eps", 1e-5)))
        elif j.get("want_stats", False):
            ws = torch.empty(cout * 2 * nblk + 2 * cout, dtype=torch.float32, device=x.device)
            scale, shift, stats = ws[:cout], ws[cout:2 * cout], ws[2 * cout:]
            a.stats = ptr(stats)
            finalize.append((stats, cout, scale, shift, float(j.get("eps", 1e-5))))
            outs.append(Pending(y, scale, shift))
        else:
            a.stats = None
            outs.append(y)
    rc = lib().ramp_conv2d_nhwc_multi(arr, len(jobs), H, W, Cin, kh, stride, code, stream())
    if rc == _lib.RAMP_EUNSUPPORTED:
        for tl in written:
            tl.out = None
        if use8:
            return conv2d_towers(jobs, half, fp8=False)      # a layer shape without an fp8 instantiation: f16 MFMA
        return single()
    check(rc, "ramp_conv2d_nhwc_multi")
    for tl in written:
        tl.written = True
    for stats, cout, scale, shift, eps in finalize:
        check(lib().ramp_in_stats_finalize(ptr(stats), nblk, cout, float(OH * OW), eps, ptr(scale), ptr(shift),
                                           stream()), "ramp_in_stats_finalize")
    return outs


def materialize(p):
    """relu(norm(raw))"""
    out = torch.empty_like(p.raw)
    fn = lib().ramp_affine_relu_f16 if p.raw.dtype == torch.float16 else lib().ramp_affine_relu
    check(fn(ptr(p.raw), ptr(p.scale), ptr(p.shift), ptr(out), p.raw.numel(), p.raw.shape[-1], stream()),
          "ramp_affine_relu")
    return out


def norm_add_relu(y, skip, fuse=True):
    """relu(skip' + relu(norm(y)));  skip is a tensor or a Pending (its norm, with or without ReLU, applied here).  Returns
    the tensor, or (fp16 towers in accumulator mode) a Tail the next layer's conv kernel evaluates while loading"""
    if isinstance(skip, Tail):
        skip = skip.tensor()
    if fuse and _TAIL_FUSE:
        t = Tail(y, skip)
        if t.fusable():
            return t
    out = torch.empty_like(y.raw)
    sp = isinstance(skip, Pending)
    if (y.raw.dtype == torch.float16 and y.acc is not None and y._scale is None and y.raw.shape[-1] <= 128
            and (not sp or (skip.acc is not None and skip._scale is None))):
        check(lib().ramp_norm_add_relu_f16_acc(ptr(y.raw), ptr(y.acc), y.count, y.eps, ptr(skip.raw if sp else skip),
                                               ptr(skip.acc) if sp else None, skip.count if sp else 0.0,
                                               skip.eps if sp else 0.0, ptr(out), y.raw.numel(), y.raw.shape[-1],
                                               int(sp and skip.relu), stream()), "ramp_norm_add_relu_f16_acc")
        return out
    s_raw, ss, hs = (skip.raw, skip.scale, skip.shift) if isinstance(skip, Pending) else (skip, None, None)
    if y.raw.dtype == torch.float16:
        check(lib().ramp_norm_add_relu_f16(ptr(y.raw), ptr(y.scale), ptr(y.shift), ptr(s_raw), ptr(ss), ptr(hs),
                                           ptr(out), y.raw.numel(), y.raw.shape[-1],
                                           int(isinstance(skip, Pending) and skip.relu), stream()),
              "ramp_norm_add_relu_f16")
        return out
    if isinstance(skip, Pending) and skip.relu:
        s_raw, ss, hs = materialize(skip), None, None
    check(lib().ramp_norm_add_relu(ptr(y.raw), ptr(y.scale), ptr(y.shift), ptr(s_raw), ptr(ss), ptr(hs), ptr(out),
                                   y.raw.numel(), y.raw.shape[-1], stream()), "ramp_norm_add_relu")
    return out


# --------------------------------------------------------------------------- towers
def _res_blocks(blks, xs, norms, half, fp8=False):
    """reference ResidualBlock.forward (extractor.py:49-57) for the same block of every tower, one launch per
    conv for all of them.  xs[t]: tensor, or (norm towers, first block) the Pending relu(norm(conv1))."""
    job = lambda t, x, conv, **k: dict(x=x, conv=conv, want_stats=norms[t], **k)
    T = range(len(blks))
    skips = list(xs)
    if blks[0].downsample is not None:
        skips = conv2d_towers([job(t, xs[t], blks[t].downsample[0]) for t in T], half, fp8)
        for t in T:
            if norms[t]:
                skips[t].relu = False            # norm3 has no ReLU behind it
    y = conv2d_towers([job(t, xs[t], blks[t].conv1, relu=not norms[t]) for t in T], half, fp8)
    # plain towers: relu(skip + relu(conv2(y))) in the conv's epilogue
    y = conv2d_towers([job(t, y[t], blks[t].conv2, relu=not norms[t], res=None if norms[t] else skips[t])
                       for t in T], half, fp8)
    return [norm_add_relu(y[t], skips[t]) if norms[t] else y[t] for t in T]


def _tower_norms(encs):
    norms = [isinstance(e.norm1, nn.InstanceNorm2d) for e in encs]
    for e, n in zip(encs, norms):
        assert n or (isinstance(e.norm1, nn.Sequential) and len(e.norm1) == 0)
    return norms


def _first_layer(encs, x, norms, half):
    xs = conv2d_towers([dict(x=x, conv=e.conv1, want_stats=n, relu=not n, eps=e.norm1.eps if n else 1e-5)
                        for e, n in zip(encs, norms)], half)
    if not half:                                  # fp32 tail kernel takes a materialised skip
        xs = [materialize(v) if isinstance(v, Pending) else v for v in xs]
    return xs


def basic_encoder4_towers(encs, x, out_scale=1.0, half=False, fp8=False):
    """BasicEncoder4._forward of every tower in ``encs`` on one NHWC image x [H,W,Cin_padded] -> [H/4,W/4,out] each
    (``half``: fp16 storage + fp16 MFMA after the first layer's fp32 input).  relu(norm1(conv1)) is never
    materialised: layer1's first conv applies it while loading, the block's tail while adding the skip."""
    norms = _tower_norms(encs)
    _scope.cur = (_scope.prepared or _AccArena(x.device)) if (half and _IN_ACC and any(norms)) else None
    _scope.prepared = None
    try:
        xs = _first_layer(encs, x, norms, half)
        for li in ("layer1", "layer2"):
            for b in range(2):
                xs = _res_blocks([getattr(e, li)[b] for e in encs], xs, norms, half, fp8)
        return conv2d_towers([dict(x=xs[t], conv=e.conv2, out_scale=out_scale, keep=False) for t, e in enumerate(encs)], half, fp8)
    finally:
        _scope.cur = None


def basic_encoder4(enc, x, out_scale=1.0, half=False):
    return basic_encoder4_towers([enc], x, out_scale, half)[0]


_MS_PAIR = True          # (False: torch.cat copies -- round 4's A/B)


def multiscale_encoder4_towers(encs, x, x2, x4, out_scale=1.0, half=False, fp8=False):
    """MultiScaleBasicEncoder4.forward (reference extractor.py:288-311) of every tower on NHWC inputs: x [H,W,16],
    x2 [H/2,W/2,32] and x4 [H/4,W/4,64] (the three super-states) -> [H/4,W/4,out].  The channel
    concatenations are the only non-conv steps; layer2/conv2 are unused, as upstream."""
    norms = _tower_norms(encs)
    _scope.cur = _AccArena(x.device) if (half and _IN_ACC and any(norms)) else None
    try:
        xs = _first_layer(encs, x, norms, half)
        for b in range(2):
            xs = _res_blocks([e.layer1[b] for e in encs], xs, norms, half, fp8)
        # the channel concatenations: two-source inputs of the next layer (fp16 towers; ramp_conv_job.x2), else copies
        two = half and _MS_PAIR
        xs = [v.tensor() if isinstance(v, Tail) else v for v in xs]           # (a two-source input takes tensors)
        x2 = x2.to(xs[0].dtype)
        xs = [Pair(v, x2) if two else torch.cat((v, x2), dim=-1) for v in xs]
        for b in range(2):
            xs = _res_blocks([e.layer3[b] for e in encs], xs, norms, half, fp8)
        xs = [v.tensor() if isinstance(v, Tail) else v for v in xs]
        x4 = x4.to(xs[0].dtype)
        xs = [Pair(v, x4) if two else torch.cat((v, x4), dim=-1) for v in xs]
        return conv2d_towers([dict(x=xs[t], conv=e.conv3, out_scale=out_scale) for t, e in enumerate(encs)], half, fp8)
    finally:
        _scope.cur = None


def multiscale_encoder4(enc, x, x2, x4, out_scale=1.0, half=False):
    return multiscale_encoder4_towers([enc], x, x2, x4, out_scale, half)[0]


# ------------------------------------------------------------------ LSTM / super-state
class LstmState:
    """recurrent state of the SingleScale front end.  h_*, c_*: tile-major [ceil(HW/16)][16 px][4 q][4 t], unit = 4 t + q
    (the MFMA kernel's layout: a lane's four operands are 16 contiguous bytes; unit 15 is padding; held here as
    [tiles, 16, 16]); ss: channels-last [HW, 16]."""
    __slots__ = ("h_ev", "c_ev", "h_im", "c_im", "ss", "flags", "fresh", "HW", "arena")

    def __init__(self, HW, device):
        z = lambda *s: torch.zeros(*s, dtype=torch.float32, device=device)
        nt = (HW + 15) // 16
        self.h_ev, self.c_ev, self.h_im, self.c_im = z(nt, 16, 16), z(nt, 16, 16), z(nt, 16, 16), z(nt, 16, 16)
        self.ss = z(HW, 16)
        self.flags = torch.zeros(2, 1024, dtype=torch.int32, device=device)     # per-workgroup results (ramp_any_nonzero_blocks)
        self.fresh = True
        self.HW = HW
        self.arena = None      # the tower pass's InstanceNorm accumulators (zeroed by the presence test's launch)

    def rows(self, name):
        """state `name` as [HW, 15] rows (pixel-major), for inspection / tests"""
        t = getattr(self, name)                     # [tile][16 px][4 q][4 t], unit = 4 t + q
        return t.view(-1, 16, 4, 4).permute(0, 1, 3, 2).reshape(-1, 16)[:self.HW, :15]


def lstm_superstate_step(enc, ev, im, st, arena_for_towers=False):
    """ev [5,H,W], im [3,H,W] contiguous fp32; updates st in place, returns the super-state
    as an NHWC16 tensor [H,W,16] (a view of st.ss)"""
    w = pack_lstm_mfma(enc)
    H, W = ev.shape[-2:]
    # the presence test is the front end's first launch: it also zeroes the InstanceNorm accumulators of the tower pass
    # that follows (a persistent arena of this state; its memset launch was 4.8 us + a boundary of every front end)
    arena = None
    if arena_for_towers and _IN_ACC:
        if st.arena is None:
            st.arena = _AccArena(ev.device, zeroed=False)
        arena = st.arena
        arena.used = 0
    nblk = lib().ramp_any_nonzero_blocks_clear(ptr(ev), ev.numel(), ptr(im), im.numel(), ptr(st.flags),
                                               ptr(arena.buf) if arena else None, arena.buf.numel() * 8 if arena else 0,
                                               stream())
    if nblk <= 0:
        check(nblk or -1, "ramp_any_nonzero_blocks_clear")
    _scope.prepared = arena
    has = 0 if st.fresh else 1
    check(lib().ramp_lstm_superstate_blocks(ptr(ev), ptr(im), ptr(st.h_ev), ptr(st.c_ev), ptr(st.h_im),
                                            ptr(st.c_im), ptr(st.ss), ptr(w), ptr(st.flags), nblk, H * W, has, has,
                                            stream()), "ramp_lstm_superstate_blocks")
    st.fresh = False
    return st.ss.view(H, W, 16)


# ----------------------------------------------------------- MultiScale front end
class MsState:
    """super-state of one scale: [Hs*Ws, D] channels-last rows"""
    __slots__ = ("s", "s16", "fresh", "Hs", "Ws", "D")

    def __init__(self, H, W, scale, device):
        k, pad = (scale + 1, 1) if scale > 1 else (1, 0)
        self.Hs, self.Ws, self.D = (H + 2 * pad - k) // scale + 1, (W + 2 * pad - k) // scale + 1, 16 * scale
        self.s = torch.zeros(self.Hs * self.Ws, self.D, dtype=torch.float32, device=device)
        self.s16 = None              # fp16 copy of the state for the fp16 towers (written by the MFMA kernel)
        self.fresh = True


def pack_ms_scale(enc, k):
    """the 12 weight arrays of ramp_ms_lstm_superstate for scale index k (see include/ramp_hip.h)"""
    ev, im = enc.ev_encoders[k], enc.im_encoders[k]
    me, mi = enc.super_state_ev_encoder[k].encoder, enc.super_state_im_encoders[k].encoder
    params = [ev.conv_1.weight, ev.conv_1.bias, im.conv_1.weight, im.conv_1.bias,
              ev.convlstm.weight_ih_l0, ev.convlstm.bias_ih_l0, ev.convlstm.bias_hh_l0,
              im.convlstm.weight_ih_l0, im.convlstm.bias_ih_l0, im.convlstm.bias_hh_l0,
              me.weight, me.bias, mi.weight, mi.bias]
    key = tuple((q.data_ptr(), q._version) for q in params)
    hit = _cache(enc).get(("ms", k))
    if hit is not None and hit[0] == key:
        return hit[1]
    f = lambda t: t.detach().float().contiguous()
    d = me.out_channels
    arrs = [f(ev.conv_1.weight), f(ev.conv_1.bias), f(im.conv_1.weight), f(im.conv_1.bias),
            f(ev.convlstm.weight_ih_l0), f(ev.convlstm.bias_ih_l0 + ev.convlstm.bias_hh_l0),
            f(im.convlstm.weight_ih_l0), f(im.convlstm.bias_ih_l0 + im.convlstm.bias_hh_l0),
            f(me.weight.view(d, 2 * d).t()), f(me.bias), f(mi.weight.view(d, 2 * d).t()), f(mi.bias)]
    import ctypes
    ptrs = (ctypes.c_void_p * 12)(*[a.data_ptr() for a in arrs])
    _cache(enc)[("ms", k)] = (key, (arrs, ptrs))
    return arrs, ptrs


def pack_ms_scale_mfma(enc, k):
    """(wfrag [nfrag, 64], wsmall) of ramp_ms_lstm_superstate_mfma for scale index k: the A operand of every
    v_mfma_f32_16x16x4_f32 as one float per lane (lane l: row i = l & 15, K column kq = l >> 4), in the order the kernel
    walks them (csrc/conv.hip::ms_lstm_superstate_mfma_kernel: gates ev [t][i,g,o][2 K steps], gates im [t][i,g,o], mix ev
    [n][K step], mix im), and the small arrays (conv_1 weights / biases, gate biases [i,g,o][D] x 2, mix biases x 2)."""
    ev, im = enc.ev_encoders[k], enc.im_encoders[k]
    me, mi = enc.super_state_ev_encoder[k].encoder, enc.super_state_im_encoders[k].encoder
    params = [ev.conv_1.weight, ev.conv_1.bias, im.conv_1.weight, im.conv_1.bias,
              ev.convlstm.weight_ih_l0, ev.convlstm.bias_ih_l0, ev.convlstm.bias_hh_l0,
              im.convlstm.weight_ih_l0, im.convlstm.bias_ih_l0, im.convlstm.bias_hh_l0,
              me.weight, me.bias, mi.weight, mi.bias]
    key = tuple((q.data_ptr(), q._version) for q in params)
    hit = _cache(enc).get(("ms_mfma", k))
    if hit is not None and hit[0] == key:
        return hit[1]
    dev = me.weight.device
    f = lambda t: t.detach().float().cpu()
    D = me.out_channels
    NG = D // 16
    lane = torch.arange(64)
    i, kq = lane & 15, lane >> 4
    gates = (0, 2, 3)                                        # i, g, o rows of torch's (i, f, g, o) order
    frags = []
    for lstm, C, steps in ((ev.convlstm, 5, 2), (im.convlstm, 3, 1)):
        Wih = f(lstm.weight_ih_l0)                           # [4D, C]
        Wp = torch.zeros(4 * D, 4 * steps)
        Wp[:, :C] = Wih
        for t in range(NG):
            for g in gates:
                for ks in range(steps):
                    frags.append(Wp[g * D + 16 * t + i, 4 * ks + kq])
    for mix in (me, mi):
        Wm = f(mix.weight).view(D, 2 * D)
        for n in range(NG):
            for half in range(2):
                for t in range(NG):
                    for r in range(4):
                        frags.append(Wm[16 * n + i, half * D + 16 * t + 4 * kq + r])
    wfrag = torch.stack(frags, 0).contiguous().to(dev)
    gb = lambda lstm: (f(lstm.bias_ih_l0) + f(lstm.bias_hh_l0)).view(4, D)[list(gates)].reshape(-1)
    wsmall = torch.cat([f(ev.conv_1.weight).reshape(-1), f(ev.conv_1.bias), f(im.conv_1.weight).reshape(-1),
                        f(im.conv_1.bias), gb(ev.convlstm), gb(im.convlstm), f(me.bias), f(mi.bias)]).contiguous().to(dev)
    assert wfrag.shape[0] == NG * 9 + 16 * NG * NG
    _cache(enc)[("ms_mfma", k)] = (key, (wfrag, wsmall))
    return wfrag, wsmall


_MS_MFMA = True          # (False: the fp32 VALU kernel -- round 4's A/B)


def ms_lstm_superstate_step(enc, k, ev, im, st, use_im, want_half=False):
    """ev [5,H,W], im [3,H,W] contiguous fp32; advances scale k's super-state in place and returns
    it as an NHWC tensor [Hs, Ws, D] (a view of st.s; with want_half the kernel's fp16 copy of it)"""
    H, W = ev.shape[-2:]
    if _MS_MFMA and (enc.scales[k] == 1 or st.Ws % 16 == 0):
        wfrag, wsmall = pack_ms_scale_mfma(enc, k)
        s16 = None
        if want_half:
            if st.s16 is None:
                st.s16 = torch.empty(st.Hs * st.Ws, st.D, dtype=torch.float16, device=st.s.device)
            s16 = st.s16
        check(lib().ramp_ms_lstm_superstate_mfma(ptr(ev), ptr(im), ptr(wfrag), ptr(wsmall), ptr(st.s), ptr(s16), H, W,
                                                 enc.scales[k], 0 if st.fresh else 1, int(bool(use_im)), stream()),
              "ramp_ms_lstm_superstate_mfma")
        st.fresh = False
        return (s16 if want_half else st.s).view(st.Hs, st.Ws, st.D)
    _, ptrs = pack_ms_scale(enc, k)
    check(lib().ramp_ms_lstm_superstate(ptr(ev), ptr(im), ptrs, ptr(st.s), H, W, enc.scales[k],
                                        0 if st.fresh else 1, int(bool(use_im)), stream()),
          "ramp_ms_lstm_superstate")
    st.fresh = False
    return st.s.view(st.Hs, st.Ws, st.D)
