"""HIP back-end of the RAMP encoder: fused per-pixel LSTM / super-state kernel and the
implicit-GEMM MFMA conv towers (csrc/conv.hip).  Activations are NHWC float32 tensors
[H, W, C]; InstanceNorm + ReLU are never materialised on their own -- they ride on the
consumer conv's load path (``pre``) or on the residual-block tail kernel.
"""
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
    mode "f32": KC=16, CPL=4 fp32;  "f16": KC=32, CPL=8 half;  "f16_first": KC=16, CPL=4 half"""
    w = conv.weight
    key = (mode, w._version, w.device, w.data_ptr())
    hit = _cache(conv).get(mode)
    if hit is not None and hit[0] == key:
        return hit[1], hit[2]
    kc, cpl = (32, 8) if mode == "f16" else (16, 4)
    cout, cin, kh, kw = w.shape
    cin_p = (cin + kc - 1) // kc * kc
    wp = torch.zeros(cout, cin_p, kh, kw, dtype=torch.float32, device=w.device)
    wp[:, :cin] = w.detach().float()
    t = wp.permute(2, 3, 1, 0).reshape(kh * kw, cin_p // kc, 4, cpl, cout // 16, 16)   # tap, ch, q, s, nt, j
    t = t.permute(0, 1, 4, 2, 5, 3).contiguous()                                        # tap, ch, nt, q, j, s
    if mode != "f32":
        t = t.half()
    bias = conv.bias.detach().float().contiguous() if conv.bias is not None else None
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
class Pending:
    """a raw conv output whose InstanceNorm(+ReLU) has not been applied yet"""
    __slots__ = ("raw", "scale", "shift")

    def __init__(self, raw, scale, shift):
        self.raw, self.scale, self.shift = raw, scale, shift


def conv2d(x, conv, pre=None, res=None, relu=False, want_stats=False, out_scale=1.0, eps=1e-5, half=False,
           direct=False):
    """x [H,W,Cin] NHWC (or a Pending: normalise+ReLU on load).  fp32 in/out, or with ``half``:
    half out and half in (fp32 in allowed for the 16-channel first layer).  Returns y [OH,OW,Cout],
    or Pending(y, scale, shift) when want_stats (InstanceNorm statistics of y, always fp32)."""
    if isinstance(x, Pending):
        pre, x = (x.scale, x.shift), x.raw
    H, W, Cin = x.shape
    if not half:
        mode, code, odt = "f32", RAMP_F32, torch.float32
    elif x.dtype == torch.float32:
        mode, code, odt = "f16_first", _lib.RAMP_F16 | _lib.RAMP_IN_F32, torch.float16
    else:
        mode, code, odt = "f16", _lib.RAMP_F16, torch.float16
    if direct and half:
        code |= _lib.RAMP_CONV_DIRECT        # the one-round-trip-per-tap kernel (A/B test of the LDS-tiled one)
    wpk, bias = pack_conv_weight(conv, mode)
    cout, _, kh, kw = conv.weight.shape
    stride = conv.stride[0]
    kc = 32 if mode == "f16" else 16
    assert Cin == wpk.shape[1] * kc and conv.padding[0] == kh // 2 and x.is_contiguous()
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


def materialize(p):
    """relu(norm(raw))"""
    out = torch.empty_like(p.raw)
    fn = lib().ramp_affine_relu_f16 if p.raw.dtype == torch.float16 else lib().ramp_affine_relu
    check(fn(ptr(p.raw), ptr(p.scale), ptr(p.shift), ptr(out), p.raw.numel(), p.raw.shape[-1], stream()),
          "ramp_affine_relu")
    return out


def norm_add_relu(y, skip):
    """relu(skip' + relu(norm(y)));  skip is a tensor or a Pending (norm, no ReLU)"""
    out = torch.empty_like(y.raw)
    s_raw, ss, hs = (skip.raw, skip.scale, skip.shift) if isinstance(skip, Pending) else (skip, None, None)
    fn = lib().ramp_norm_add_relu_f16 if y.raw.dtype == torch.float16 else lib().ramp_norm_add_relu
    check(fn(ptr(y.raw), ptr(y.scale), ptr(y.shift), ptr(s_raw), ptr(ss), ptr(hs), ptr(out), y.raw.numel(),
             y.raw.shape[-1], stream()), "ramp_norm_add_relu")
    return out


# --------------------------------------------------------------------------- towers
def _res_block(blk, x, norm, half):
    """reference ResidualBlock.forward (extractor.py:49-57)"""
    if norm:
        y = conv2d(x, blk.conv1, want_stats=True, half=half)
        y = conv2d(y, blk.conv2, want_stats=True, half=half)
        skip = x if blk.downsample is None else conv2d(x, blk.downsample[0], want_stats=True, half=half)
        return norm_add_relu(y, skip)
    y = conv2d(x, blk.conv1, relu=True, half=half)
    skip = x if blk.downsample is None else conv2d(x, blk.downsample[0], half=half)
    return conv2d(y, blk.conv2, res=skip, relu=True, half=half)       # relu(skip + relu(conv2(y)))


def basic_encoder4(enc, x, out_scale=1.0, half=False):
    """BasicEncoder4._forward on one NHWC image x [H,W,Cin_padded] -> [H/4,W/4,out]
    (``half``: fp16 storage + fp16 MFMA after the first layer's fp32 input)"""
    norm = isinstance(enc.norm1, nn.InstanceNorm2d)
    if norm:
        x = materialize(conv2d(x, enc.conv1, want_stats=True, eps=enc.norm1.eps, half=half))
    else:
        assert isinstance(enc.norm1, nn.Sequential) and len(enc.norm1) == 0
        x = conv2d(x, enc.conv1, relu=True, half=half)
    for blk in enc.layer1:
        x = _res_block(blk, x, norm, half)
    for blk in enc.layer2:
        x = _res_block(blk, x, norm, half)
    return conv2d(x, enc.conv2, out_scale=out_scale, half=half)


def multiscale_encoder4(enc, x, x2, x4, out_scale=1.0, half=False):
    """MultiScaleBasicEncoder4.forward (reference extractor.py:288-311) on NHWC inputs: x [H,W,16],
    x2 [H/2,W/2,32] and x4 [H/4,W/4,64] (the three super-states) -> [H/4,W/4,out].  The channel
    concatenations are the only non-conv steps; layer2/conv2 are unused, as upstream."""
    norm = isinstance(enc.norm1, nn.InstanceNorm2d)
    if norm:
        x = materialize(conv2d(x, enc.conv1, want_stats=True, eps=enc.norm1.eps, half=half))
    else:
        assert isinstance(enc.norm1, nn.Sequential) and len(enc.norm1) == 0
        x = conv2d(x, enc.conv1, relu=True, half=half)
    for blk in enc.layer1:
        x = _res_block(blk, x, norm, half)
    x = torch.cat((x, x2.to(x.dtype)), dim=-1)
    for blk in enc.layer3:
        x = _res_block(blk, x, norm, half)
    x = torch.cat((x, x4.to(x.dtype)), dim=-1)
    return conv2d(x, enc.conv3, out_scale=out_scale, half=half)


# ------------------------------------------------------------------ LSTM / super-state
class LstmState:
    """recurrent state of the SingleScale front end.  h_*, c_*: tile-major [ceil(HW/16), 16 units,
    16 px] (the MFMA kernel's layout; unit 15 is padding); ss: channels-last [HW, 16]."""
    __slots__ = ("h_ev", "c_ev", "h_im", "c_im", "ss", "flags", "fresh", "HW")

    def __init__(self, HW, device):
        z = lambda *s: torch.zeros(*s, dtype=torch.float32, device=device)
        nt = (HW + 15) // 16
        self.h_ev, self.c_ev, self.h_im, self.c_im = z(nt, 16, 16), z(nt, 16, 16), z(nt, 16, 16), z(nt, 16, 16)
        self.ss = z(HW, 16)
        self.flags = torch.zeros(2, dtype=torch.int32, device=device)
        self.fresh = True
        self.HW = HW

    def rows(self, name):
        """state `name` as [HW, 15] rows (pixel-major), for inspection / tests"""
        t = getattr(self, name)
        return t.permute(0, 2, 1).reshape(-1, 16)[:self.HW, :15]


def lstm_superstate_step(enc, ev, im, st):
    """ev [5,H,W], im [3,H,W] contiguous fp32; updates st in place, returns the super-state
    as an NHWC16 tensor [H,W,16] (a view of st.ss)"""
    w = pack_lstm_mfma(enc)
    H, W = ev.shape[-2:]
    check(lib().ramp_any_nonzero(ptr(ev), ev.numel(), ptr(im), im.numel(), ptr(st.flags), stream()),
          "ramp_any_nonzero")
    has = 0 if st.fresh else 1
    check(lib().ramp_lstm_superstate_tiled(ptr(ev), ptr(im), ptr(st.h_ev), ptr(st.c_ev), ptr(st.h_im),
                                           ptr(st.c_im), ptr(st.ss), ptr(w), ptr(st.flags), H * W, has, has,
                                           stream()), "ramp_lstm_superstate_tiled")
    st.fresh = False
    return st.ss.view(H, W, 16)


# ----------------------------------------------------------- MultiScale front end
class MsState:
    """super-state of one scale: [Hs*Ws, D] channels-last rows"""
    __slots__ = ("s", "fresh", "Hs", "Ws", "D")

    def __init__(self, H, W, scale, device):
        k, pad = (scale + 1, 1) if scale > 1 else (1, 0)
        self.Hs, self.Ws, self.D = (H + 2 * pad - k) // scale + 1, (W + 2 * pad - k) // scale + 1, 16 * scale
        self.s = torch.zeros(self.Hs * self.Ws, self.D, dtype=torch.float32, device=device)
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


def ms_lstm_superstate_step(enc, k, ev, im, st, use_im):
    """ev [5,H,W], im [3,H,W] contiguous fp32; advances scale k's super-state in place and returns
    it as an NHWC tensor [Hs, Ws, D] (a view of st.s)"""
    _, ptrs = pack_ms_scale(enc, k)
    H, W = ev.shape[-2:]
    check(lib().ramp_ms_lstm_superstate(ptr(ev), ptr(im), ptrs, ptr(st.s), H, W, enc.scales[k],
                                        0 if st.fresh else 1, int(bool(use_im)), stream()),
          "ramp_ms_lstm_superstate")
    st.fresh = False
    return st.s.view(st.Hs, st.Ws, st.D)
