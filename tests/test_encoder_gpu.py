"""GPU tests of the encoder kernels (csrc/conv.hip): the MFMA implicit-GEMM conv, the fused
LSTM/super-state kernel and the InstanceNorm plumbing against plain PyTorch fp32 on the same
device (the numerics reference for a floating-point kernel), plus the whole SingleScale encoder
HIP path against the ATen/MIOpen path."""
import numpy as np
import pytest
import torch
import torch.nn as nn
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _ref_conv(x_nhwc, conv):
    x = x_nhwc.permute(2, 0, 1)[None]
    w = conv.weight
    if x.shape[1] != w.shape[1]:
        x = x[:, :w.shape[1]]
    return F.conv2d(x, w, conv.bias, conv.stride, conv.padding)[0].permute(1, 2, 0)


@pytest.mark.parametrize("cin,cout,k,stride,hw", [(16, 32, 7, 2, (48, 64)), (32, 32, 3, 1, (37, 53)),
                                                   (32, 64, 3, 2, (40, 56)), (64, 64, 3, 1, (20, 28)),
                                                   (32, 64, 1, 2, (40, 56)), (64, 128, 1, 1, (20, 28)),
                                                   (64, 384, 1, 1, (21, 29))])
def test_conv_mfma_matches_torch(cin, cout, k, stride, hw):
    from rampvo_amd import conv_hip
    torch.manual_seed(0)
    real_cin = 15 if cin == 16 else cin
    conv = nn.Conv2d(real_cin, cout, k, stride=stride, padding=k // 2).cuda()
    x = torch.randn(hw[0], hw[1], cin, device="cuda")
    if cin == 16:
        x[..., 15] = 0
    with torch.no_grad():
        ref = _ref_conv(x, conv)
        y = conv_hip.conv2d(x, conv)
        assert y.shape == ref.shape
        tol = 2e-5 * float(ref.abs().max()) * (1 + (real_cin * k * k) ** 0.5 / 8)
        assert float((y - ref).abs().max()) <= tol
        # fused prologue (normalise + relu on load), epilogue relu / residual / scale, statistics
        sc, sh = torch.rand(cin, device="cuda") + 0.5, torch.randn(cin, device="cuda") * 0.1
        xin = F.relu(x * sc + sh)
        if cin == 16:
            xin[..., 15] = 0
        res = torch.randn(ref.shape, device="cuda")      # contiguous NHWC
        ref2 = F.relu(res + F.relu(_ref_conv(xin, conv))) * 0.25
        y2 = conv_hip.conv2d(x, conv, pre=(sc, sh), res=res, relu=True, out_scale=0.25)
        assert float((y2 - ref2).abs().max()) <= tol
        p = conv_hip.conv2d(x, conv, want_stats=True)
        mean = ref.reshape(-1, cout).mean(0)
        var = ref.reshape(-1, cout).var(0, unbiased=False)
        assert float((p.scale - (var + 1e-5).rsqrt()).abs().max()) <= 1e-3 * float((var + 1e-5).rsqrt().max())
        assert float((p.shift + mean * (var + 1e-5).rsqrt()).abs().max()) <= 2e-3
        refn = F.relu(F.instance_norm(ref.permute(2, 0, 1)[None], eps=1e-5))[0].permute(1, 2, 0)
        assert float((conv_hip.materialize(p) - refn).abs().max()) <= 2e-3


@pytest.mark.parametrize("cin,cout,k,stride,hw", [(32, 32, 3, 1, (24, 40)), (32, 64, 3, 2, (24, 40)), (64, 64, 3, 1, (25, 33)),
                                                  (16, 32, 7, 2, (48, 64)), (64, 64, 3, 1, (120, 160)), (16, 32, 7, 2, (97, 131))])
def test_conv_x3_against_fp64_and_the_exact_product_kernel(cin, cout, k, stride, hw):
    """fp32 towers, round 6: csrc/conv.hip::conv_x3_kernel -- every operand split into three fp16 numbers (33 bits), the six
    f16 MFMA products of order <= 2 into three fp32 accumulators: exact-class products, only sums are rounded, as in the
    reference -- against a float64 convolution (error relative to sum |x||w|, the scale of the rounding: <= 2e-6 and no
    worse than 1.5x the exact-product f32 MFMA kernel it replaces) and against that kernel; the InstanceNorm statistics it
    emits; prologue affine + ReLU, bias, ReLU, residual, out_scale; inputs with entries below the fp16 normal range and large
    ones"""
    from rampvo_amd import conv_hip
    from rampvo_amd._lib import RAMP_F32, RAMP_CONV_X3, lib
    torch.manual_seed(5)
    H, W = hw
    conv = nn.Conv2d(cin, cout, k, stride, k // 2).cuda()
    x = torch.randn(H, W, cin, device="cuda")
    x[::3, ::5] *= 1e-6                                           # below the fp16 normal range
    x[1::4, 2::7] *= 300.0
    sc = torch.rand(cin, device="cuda") + 0.5
    sh = torch.randn(cin, device="cuda") * 0.2
    OH, OW = (H + 2 * (k // 2) - k) // stride + 1, (W + 2 * (k // 2) - k) // stride + 1
    res = torch.randn(OH, OW, cout, device="cuda")
    assert lib().ramp_conv2d_stats_blocks(H, W, cin, cout, k, stride, RAMP_F32 | RAMP_CONV_X3) > 0
    out = {}
    with torch.no_grad():
        for x3 in (True, False):
            conv_hip.X3 = x3
            try:
                y = conv_hip.conv2d(x, conv, pre=(sc, sh), res=res, relu=True, out_scale=0.25)
                pend = conv_hip.conv2d(x, conv, want_stats=True)
                out[x3] = (y, pend.raw, pend.scale, pend.shift)
            finally:
                conv_hip.X3 = True
        xd = x.double().permute(2, 0, 1)[None]
        w, b = conv.weight.double(), conv.bias.double()
        raw = F.conv2d(xd, w, b, stride, k // 2)[0].permute(1, 2, 0)
        mag = F.conv2d(xd.abs(), w.abs(), b.abs(), stride, k // 2)[0].permute(1, 2, 0)
        e3 = float(((out[True][1].double() - raw).abs() / mag).max())
        e1 = float(((out[False][1].double() - raw).abs() / mag).max())
        print("max error / sum |x||w|: split %.2e, exact products %.2e" % (e3, e1))
        assert e3 <= 2e-6 and e1 <= 2e-6 and e3 <= 1.5 * e1 + 1e-8, (e3, e1)
        xa = torch.relu(x.double() * sc.double() + sh.double()).permute(2, 0, 1)[None]
        full = torch.relu(torch.relu(F.conv2d(xa, w, b, stride, k // 2)[0].permute(1, 2, 0)) + res.double()) * 0.25
        magf = F.conv2d(xa.abs(), w.abs(), b.abs(), stride, k // 2)[0].permute(1, 2, 0) * 0.25 + res.double().abs() * 0.25
        assert float(((out[True][0].double() - full).abs() / magf).max()) <= 2e-6
        assert (out[True][0] - out[False][0]).abs().max() <= 1e-5 * float(out[False][0].abs().max())
        # InstanceNorm statistics of the raw output (scale = rsqrt(var + eps), shift = -mean scale)
        mean, var = raw.mean((0, 1)), raw.var((0, 1), unbiased=False)
        assert torch.allclose(out[True][2].double(), 1.0 / torch.sqrt(var + 1e-5), rtol=1e-4)
        assert torch.allclose(out[True][3].double(), -mean / torch.sqrt(var + 1e-5), rtol=1e-4, atol=1e-5)


@pytest.mark.parametrize("cin,cout,k,stride,hw,first", [(16, 32, 7, 2, (48, 64), True), (32, 32, 3, 1, (37, 53), False),
                                                         (32, 64, 3, 2, (40, 56), False), (64, 64, 3, 1, (20, 28), False),
                                                         (32, 64, 1, 2, (40, 56), False), (64, 384, 1, 1, (21, 29), False)])
def test_conv_mfma_f16_matches_torch(cin, cout, k, stride, hw, first):
    """fp16 storage / fp16 MFMA (fp32 accumulate) against fp32 torch on the fp16-rounded operands"""
    from rampvo_amd import conv_hip
    torch.manual_seed(2)
    real_cin = 15 if cin == 16 else cin
    conv = nn.Conv2d(real_cin, cout, k, stride=stride, padding=k // 2).cuda()
    with torch.no_grad():
        conv.weight.copy_(conv.weight.half().float())
        x = torch.randn(hw[0], hw[1], cin, device="cuda").half().float()
        if cin == 16:
            x[..., 15] = 0
        xin = x if first else x.half()
        ref = _ref_conv(x, conv)
        y = conv_hip.conv2d(xin, conv, half=True)
        assert y.dtype == torch.float16 and y.shape == ref.shape
        tol = 2e-3 * float(ref.abs().max())
        assert float((y.float() - ref).abs().max()) <= tol
        sc, sh = torch.rand(cin, device="cuda") + 0.5, torch.randn(cin, device="cuda") * 0.1
        xq = F.relu(x * sc + sh).half().float()           # the kernel rounds the normalised input to half
        if cin == 16:
            xq[..., 15] = 0
        res = torch.randn(ref.shape, device="cuda").half()
        ref2 = F.relu(res.float() + F.relu(_ref_conv(xq, conv))) * 0.25
        y2 = conv_hip.conv2d(xin, conv, pre=(sc, sh), res=res, relu=True, out_scale=0.25, half=True)
        assert float((y2.float() - ref2).abs().max()) <= 2 * tol
        p = conv_hip.conv2d(xin, conv, want_stats=True, half=True)
        refn = F.relu(F.instance_norm(ref.permute(2, 0, 1)[None], eps=1e-5))[0].permute(1, 2, 0)
        assert float((conv_hip.materialize(p).float() - refn).abs().max()) <= 1e-2


@pytest.mark.parametrize("cin,cout,k,stride,hw,first", [(16, 32, 7, 2, (48, 64), True), (32, 32, 3, 1, (37, 53), False),
                                                         (32, 64, 3, 2, (40, 56), False), (64, 64, 3, 1, (20, 28), False),
                                                         (32, 64, 1, 2, (40, 56), False), (64, 384, 1, 1, (21, 29), False),
                                                         (64, 64, 1, 2, (33, 47), False), (128, 128, 1, 1, (19, 35), False),
                                                         (32, 32, 3, 1, (240, 320), False)])
def test_conv_tiled_equals_direct(cin, cout, k, stride, hw, first):
    """the LDS-tiled fp16 kernel accumulates in the same order as the direct one: bit-identical
    outputs, and InstanceNorm statistics equal up to the (different) block partition of the sum"""
    from rampvo_amd import conv_hip
    torch.manual_seed(3)
    conv = nn.Conv2d(cin, cout, k, stride=stride, padding=k // 2).cuda()
    with torch.no_grad():
        x = torch.randn(hw[0], hw[1], cin, device="cuda")
        xin = x if first else x.half()
        sc, sh = torch.rand(cin, device="cuda") + 0.5, torch.randn(cin, device="cuda") * 0.1
        oh, ow = (hw[0] + 2 * (k // 2) - k) // stride + 1, (hw[1] + 2 * (k // 2) - k) // stride + 1
        res = torch.randn(oh, ow, cout, device="cuda").half()
        for kw in (dict(), dict(relu=True), dict(pre=(sc, sh), res=res, relu=True, out_scale=0.25)):
            a = conv_hip.conv2d(xin, conv, half=True, direct=True, **kw)
            b = conv_hip.conv2d(xin, conv, half=True, **kw)
            assert torch.equal(a, b), kw.keys()
        pa = conv_hip.conv2d(xin, conv, want_stats=True, half=True, direct=True)
        pb = conv_hip.conv2d(xin, conv, want_stats=True, half=True)
        assert torch.equal(pa.raw, pb.raw)
        assert float((pa.scale - pb.scale).abs().max()) <= 1e-5 * float(pa.scale.abs().max())
        assert float((pa.shift - pb.shift).abs().max()) <= 1e-5 * max(1.0, float(pa.shift.abs().max()))


@pytest.mark.parametrize("cin,couts,k,stride,hw,first", [(16, (32, 32), 7, 2, (96, 128), True),
                                                          (32, (32, 32), 3, 1, (37, 53), False),
                                                          (32, (64, 64), 3, 2, (40, 56), False),
                                                          (64, (64, 64), 3, 1, (120, 160), False),
                                                          (32, (64, 64), 1, 2, (40, 56), False),
                                                          (64, (128, 384), 1, 1, (21, 29), False)])
def test_conv_tower_pair_equals_single_launches(cin, couts, k, stride, hw, first):
    """two towers' layer as ONE launch (ramp_conv2d_nhwc_multi; ramp/extractor.py:233-259 runs fnet and inet on the
    same input): outputs and InstanceNorm (scale, shift) bit-identical to the per-tower launches, and stable under
    repetition"""
    from rampvo_amd import conv_hip
    torch.manual_seed(5)
    convs = [nn.Conv2d(cin, c, k, stride=stride, padding=k // 2).cuda() for c in couts]
    with torch.no_grad():
        x = torch.randn(hw[0], hw[1], cin, device="cuda")
        xin = x if first else x.half()
        sc, sh = torch.rand(cin, device="cuda") + 0.5, torch.randn(cin, device="cuda") * 0.1
        oh, ow = (hw[0] + 2 * (k // 2) - k) // stride + 1, (hw[1] + 2 * (k // 2) - k) // stride + 1
        res = torch.randn(oh, ow, couts[1], device="cuda").half()
        xa = conv_hip.Pending(xin, sc, sh) if not first else xin
        # tower 0: norm tower (statistics, no ReLU); tower 1: plain tower (ReLU + residual + scale)
        jobs = [dict(x=xa, conv=convs[0], want_stats=True, eps=1e-5),
                dict(x=xin, conv=convs[1], relu=True, res=res, out_scale=0.25)]
        pair = conv_hip.conv2d_towers(jobs, half=True)
        ref0 = conv_hip.conv2d(xa, convs[0], want_stats=True, half=True)
        ref1 = conv_hip.conv2d(xin, convs[1], relu=True, res=res, out_scale=0.25, half=True)
        assert isinstance(pair[0], conv_hip.Pending) and torch.equal(pair[0].raw, ref0.raw)
        assert torch.equal(pair[1], ref1)
        assert torch.equal(pair[0].scale, ref0.scale) and torch.equal(pair[0].shift, ref0.shift)
        for _ in range(5):
            again = conv_hip.conv2d_towers(jobs, half=True)
            assert torch.equal(again[0].scale, pair[0].scale) and torch.equal(again[0].shift, pair[0].shift)
            assert torch.equal(again[1], pair[1])


@pytest.mark.parametrize("cin,cout,k,stride,hw", [(32, 32, 3, 1, (120, 160)), (32, 64, 3, 2, (61, 83)), (64, 64, 3, 1, (60, 80)),
                                                  (64, 128, 1, 1, (30, 40))])
def test_instance_norm_accumulators_match_the_finalize_launch(cin, cout, k, stride, hw):
    """accumulator mode of the InstanceNorm statistics (csrc/conv.hip: producers add exact 2^-20 fixed-point partial
    sums to one of 8 replicas, consumers reduce them): the (scale, shift) the accumulators give (ramp_in_acc_finalize)
    against ramp_in_stats_finalize on the same layer, the consuming conv and the residual tail against the classic
    path, and bit stability of everything under repetition (integer sums: no order dependence)"""
    from rampvo_amd import conv_hip
    torch.manual_seed(11)
    conv_a = nn.Conv2d(cin, cout, k, stride=stride, padding=k // 2).cuda()
    conv_b = nn.Conv2d(cout, cout, 3, padding=1).cuda()
    dummy = nn.Conv2d(cin, cout, k, stride=stride, padding=k // 2).cuda()      # second tower: the paired launch
    with torch.no_grad():
        x = (torch.randn(hw[0], hw[1], cin, device="cuda") * 1.5 + 0.3).half()

        def run(acc):
            conv_hip.set_arena(conv_hip._AccArena(x.device) if acc else None)
            try:
                pa = conv_hip.conv2d_towers([dict(x=x, conv=conv_a, want_stats=True), dict(x=x, conv=dummy)], half=True)[0]
                assert (pa.acc is not None) == acc
                pb = conv_hip.conv2d_towers([dict(x=pa, conv=conv_b, want_stats=True), dict(x=pa.raw, conv=conv_b)], half=True)[0]
                out = conv_hip.norm_add_relu(pb, pa.raw, fuse=False)
                return pa, pb, out
            finally:
                conv_hip.set_arena(None)

        ca, cb, cout_ = run(False)
        aa, ab, aout = run(True)
        assert torch.equal(aa.raw, ca.raw)
        for acc_p, ref_p in ((aa, ca), (ab, cb)):
            sc, sh = acc_p.scale, acc_p.shift                  # (a finalize launch of its own, from the accumulators)
            assert float((sc - ref_p.scale).abs().max()) <= 2e-6 * float(ref_p.scale.abs().max())
            assert float((sh - ref_p.shift).abs().max()) <= 2e-6 * max(1.0, float(ref_p.shift.abs().max()))
        # the consumers: same values up to the rounding of (scale, shift)
        assert float((ab.raw.float() - cb.raw.float()).abs().max()) <= 2e-3 * float(cb.raw.float().abs().max())
        assert float((aout.float() - cout_.float()).abs().max()) <= 4e-3 * float(cout_.float().abs().max())
        for _ in range(3):
            ra, rb, rout = run(True)
            assert torch.equal(ra.acc, aa.acc) and (ab.acc is None or torch.equal(rb.acc, ab.acc))   # (None: a layer shape on the per-tower kernel)
            assert torch.equal(rb.raw, ab.raw) and torch.equal(rout, aout)


ENC_HIP_VS_ATEN_TOL = 4e-5  # fp32 HIP encoder against the ATen forward of the same modules, x the map's largest entry (measured 4.9e-6 / 9.4e-6)
FP8_LAYER_TOL = 2e-3        # x the output scale: what is left after both sides use the SAME e4m3-rounded operands
FP8_ENCODER_TOL = 0.2       # max abs error of fmap / imap against the f16-MFMA towers, x the map's largest entry (measured: 0.07-0.125)


@pytest.mark.parametrize("cin,couts,k,stride,hw", [(32, (32, 32), 3, 1, (37, 53)), (32, (64, 64), 3, 2, (40, 56)),
                                                   (64, (64, 64), 3, 1, (30, 44)), (32, (64, 64), 1, 2, (40, 56)),
                                                   (64, (128, 384), 1, 1, (21, 29)), (128, (128, 384), 1, 1, (19, 35))])
def test_conv_fp8_mfma_layer_against_fp32_on_rounded_operands(cin, couts, k, stride, hw):
    """BASELINE configs[4]'s "fp16 encoder on fp8 MFMA" (ramp_conv2d_nhwc_multi with RAMP_CONV_FP8): the kernel
    multiplies e4m3(x * 8) by e4m3(w * 448 / max|w|) and accumulates in fp32.  Products of e4m3 numbers are exact in
    fp32, so a plain fp32 convolution of the SAME rounded operands is the reference: what separates the two is the
    accumulation order and the fp16 rounding of the output.  Prologue (InstanceNorm + ReLU on load), bias,
    statistics, ReLU, residual and scale are the f16 kernel's own (shared epilogue)."""
    from rampvo_amd import conv_hip
    torch.manual_seed(7)
    convs = [nn.Conv2d(cin, c, k, stride=stride, padding=k // 2).cuda() for c in couts]
    e4 = torch.float8_e4m3fn
    with torch.no_grad():
        x = (torch.randn(hw[0], hw[1], cin, device="cuda") * 1.5).half()
        sc, sh = torch.rand(cin, device="cuda") + 0.5, torch.randn(cin, device="cuda") * 0.1
        oh, ow = (hw[0] + 2 * (k // 2) - k) // stride + 1, (hw[1] + 2 * (k // 2) - k) // stride + 1
        res = torch.randn(oh, ow, couts[1], device="cuda").half()
        jobs = [dict(x=conv_hip.Pending(x, sc, sh), conv=convs[0], want_stats=True),
                dict(x=x, conv=convs[1], relu=True, res=res, out_scale=0.25)]
        out = conv_hip.conv2d_towers(jobs, half=True, fp8=True)

        def ref(xin, conv):
            a = conv_hip.FP8_ACT_SCALE
            xq = (xin * a).clamp(-448, 448).to(e4).float() / a
            ws = 448.0 / float(conv.weight.abs().max())
            wq = (conv.weight.float() * ws).clamp(-448, 448).to(e4).float() / ws
            return F.conv2d(xq.permute(2, 0, 1)[None], wq, conv.bias.float(), conv.stride, conv.padding)[0].permute(1, 2, 0)
        r0 = ref(torch.relu(x.float() * sc + sh), convs[0])
        r1 = torch.relu(torch.relu(ref(x.float(), convs[1])) + res.float()) * 0.25
        assert float((out[0].raw.float() - r0).abs().max()) <= FP8_LAYER_TOL * float(r0.abs().max())
        assert float((out[1].float() - r1).abs().max()) <= FP8_LAYER_TOL * max(1.0, float(r1.abs().max()))
        mean, var = r0.mean((0, 1)), r0.var((0, 1), unbiased=False)
        assert float((out[0].scale - (var + 1e-5).rsqrt()).abs().max()) <= 2e-3 * float(out[0].scale.abs().max())
        assert float((out[0].shift + mean * (var + 1e-5).rsqrt()).abs().max()) <= 2e-3 * max(1.0, float(out[0].shift.abs().max()))
        # and the size of the step from the f16 MFMA, for the record (e4m3 carries 4 significant bits)
        f16 = conv_hip.conv2d_towers(jobs, half=True)
        print("fp8 vs f16 MFMA layer: max abs %.3g of %.3g" % (float((out[1].float() - f16[1].float()).abs().max()),
                                                             float(f16[1].float().abs().max())))


@pytest.mark.parametrize("mode", ["SingleScale", "MultiScale"])
def test_encoder_fp8_mfma_against_f16_mfma(mode):
    """the whole front end with ENCODER_FP8 against the same front end on the f16 MFMA (three frames, recurrent state
    carried): fmap / imap within FP8_ENCODER_TOL of their largest entry -- eleven layers of 4-bit-mantissa products,
    re-normalised by InstanceNorm in the fmap tower; the first layer (fp32 input) stays on the f16 MFMA"""
    from rampvo_amd.synthetic import SyntheticStream, make_network
    stream = SyntheticStream(96, 128, 3, seed=6)
    outs = {}
    with torch.no_grad():
        for fp8 in (False, True):
            net = make_network(mode)
            enc = net.patchify.encoder
            enc.mixed_precision, enc.fp8_mfma = True, fp8
            res = []
            for t in range(3):
                im, ev, _, mask = stream.frame(t)
                if mode == "SingleScale":
                    f, i, _ = enc(events=ev.cuda(), images=im.cuda(), reinit_hidden=(t == 0), out_scale=0.25)
                else:
                    f, i = enc(events=ev.cuda(), images=im.cuda(), mask=torch.ones(1, dtype=torch.bool), reinit_hidden=(t == 0),
                               out_scale=0.25)
                res.append((f.float().clone(), i.float().clone()))
            outs[fp8] = res
    worst = 0.0
    for (f0, i0), (f1, i1) in zip(outs[False], outs[True]):
        ef, ei = float((f0 - f1).abs().max()) / float(f0.abs().max()), float((i0 - i1).abs().max()) / float(i0.abs().max())
        worst = max(worst, ef, ei)
        assert ef <= FP8_ENCODER_TOL and ei <= FP8_ENCODER_TOL, (ef, ei)
    print("%s fp8 vs f16 MFMA encoder: worst relative max-abs error %.3g" % (mode, worst))


def test_singlescale_encoder_half_vs_fp32():
    from rampvo_amd.synthetic import SyntheticStream, make_network
    net = make_network("SingleScale")
    enc = net.patchify.encoder
    stream = SyntheticStream(96, 128, 3, seed=6)
    outs = {}
    with torch.no_grad():
        for half in (False, True):
            enc.mixed_precision = half
            res = []
            for t in range(3):
                im, ev, _, _ = stream.frame(t)
                f, i, _ = enc(events=ev.cuda(), images=im.cuda(), reinit_hidden=(t == 0), out_scale=0.25)
                res.append((f.float().clone(), i.float().clone()))
            outs[half] = res
    enc.mixed_precision = False
    for (f0, i0), (f1, i1) in zip(outs[False], outs[True]):
        assert float((f0 - f1).abs().max()) <= 3e-2 * float(f0.abs().max())
        assert float((i0 - i1).abs().max()) <= 3e-2 * float(i0.abs().max())


def test_lstm_superstate_kernel_matches_torch():
    from rampvo_amd import conv_hip
    from rampvo_amd.extractor import MergerLSTMsceneEncoder
    torch.manual_seed(1)
    enc = MergerLSTMsceneEncoder().cuda().eval()
    H, W = 23, 41          # H*W not a multiple of the 16-pixel MFMA tile
    st = conv_hip.LstmState(H * W, "cuda")
    h_e = c_e = h_i = c_i = s = None
    with torch.no_grad():
        for t in range(3):
            ev = torch.randn(5, H, W, device="cuda") * (0 if t == 1 else 1)      # t=1: no events
            im = torch.randn(3, H, W, device="cuda")
            out = conv_hip.lstm_superstate_step(enc, ev, im, st).clone()
            # torch reference: nn.LSTM over per-pixel sequences of length 1, conv1x1 super-state
            xe = ev.permute(1, 2, 0).reshape(-1, 1, 5)
            xi = im.permute(1, 2, 0).reshape(-1, 1, 3)
            oe, (h_e, c_e) = enc.events_convlstm(xe, None if h_e is None else (h_e, c_e))
            oi, (h_i, c_i) = enc.image_convlstm(xi, None if h_i is None else (h_i, c_i))
            if s is None:
                s = torch.zeros(15, H, W, device="cuda")
            for emb, present in ((oe, bool((ev != 0).any())), (oi, bool((im != 0).any()))):
                if present:
                    e = emb.reshape(H, W, 15).permute(2, 0, 1)
                    s = enc.superstate_encoder(torch.cat((s, e), 0))
            assert float((out[..., :15] - s.permute(1, 2, 0)).abs().max()) <= 2e-5
            assert float(out[..., 15].abs().max()) == 0.0
            assert float((st.rows("h_ev") - h_e[0]).abs().max()) <= 2e-5
            assert float((st.rows("c_im") - c_i[0]).abs().max()) <= 2e-5


def test_singlescale_encoder_hip_vs_aten():
    """fused LSTM / super-state kernel + MFMA conv towers (fp32) against the plain-PyTorch restatement of the reference
    forward (oracle/host_cpu.py, run on the GPU through ATen)"""
    from oracle import host_cpu
    from rampvo_amd.synthetic import SyntheticStream, make_network
    stream = SyntheticStream(96, 128, 3, seed=5)
    outs = {}
    with torch.no_grad():
        for backend in ("torch", "hip"):
            enc = make_network("SingleScale").patchify.encoder
            fwd = (lambda **k: host_cpu.merger_forward(enc, **k)) if backend == "torch" else enc
            res = []
            for t in range(3):
                im, ev, _, _ = stream.frame(t)
                if backend == "torch":
                    with _torch_towers():
                        f, i, _ = fwd(events=ev.cuda(), images=im.cuda(), reinit_hidden=(t == 0), out_scale=0.25)
                else:
                    f, i, _ = fwd(events=ev.cuda(), images=im.cuda(), reinit_hidden=(t == 0), out_scale=0.25)
                res.append((f.float().clone(), i.float().clone()))
            outs[backend] = res
    worst = 0.0
    for (f0, i0), (f1, i1) in zip(outs["torch"], outs["hip"]):
        assert f0.shape == f1.shape and i0.shape == i1.shape
        ef = float((f0 - f1).abs().max()) / float(f0.abs().max())
        ei = float((i0 - i1).abs().max()) / float(i0.abs().max())
        worst = max(worst, ef, ei)
        assert ef <= ENC_HIP_VS_ATEN_TOL and ei <= ENC_HIP_VS_ATEN_TOL, (ef, ei)
    print("HIP fp32 encoder vs ATen, worst max-abs / max:", worst)


def _torch_towers():
    """bind the torch forwards of the tower modules (ResidualBlock ...) for the duration of the reference run"""
    import contextlib
    from oracle import host_cpu

    @contextlib.contextmanager
    def ctx():
        saved = [(c, a, c.__dict__.get(a)) for c, a, _ in host_cpu.module_patches()]
        for c, a, f in host_cpu.module_patches():
            setattr(c, a, f)
        try:
            yield
        finally:
            for c, a, f in saved:
                setattr(c, a, f)
    return ctx()


def test_multiscale_encoder_hip_vs_aten():
    """MultiScale front end: fused conv_1 + zero-state LSTM + super-state kernel per scale and the
    MFMA towers against the plain-PyTorch restatement, including an events-only step (mask False)"""
    from oracle import host_cpu
    from rampvo_amd.synthetic import SyntheticStream, make_network
    stream = SyntheticStream(96, 128, 4, seed=6)
    masks = [True, True, False, True]
    outs = {}
    with torch.no_grad():
        for backend in ("torch", "hip"):
            enc = make_network("MultiScale").patchify.encoder
            res = []
            for t in range(4):
                im, ev, _, _ = stream.frame(t)
                kw = dict(events=ev.cuda(), images=im.cuda(), mask=torch.tensor([masks[t]]), reinit_hidden=(t == 0),
                          out_scale=0.25)
                if backend == "torch":
                    with _torch_towers():
                        f, i = host_cpu.multiscale_forward(enc, **kw)
                else:
                    f, i = enc(**kw)
                if masks[t]:
                    res.append((f.float().clone(), i.float().clone()))
            outs[backend] = res
            if backend == "hip":
                states = [s.s.clone() for s in enc._hip_state]
            else:
                ref_states = [s.clone() for s in enc.super_states]
    for a, b in zip(ref_states, states):
        assert a.shape == b.shape and float((a - b).abs().max()) <= 2e-5 * max(1.0, float(a.abs().max()))
    assert len(outs["hip"]) == 3
    worst = 0.0
    for (f0, i0), (f1, i1) in zip(outs["torch"], outs["hip"]):
        assert f0.shape == f1.shape and i0.shape == i1.shape
        ef = float((f0 - f1).abs().max()) / float(f0.abs().max())
        ei = float((i0 - i1).abs().max()) / float(i0.abs().max())
        worst = max(worst, ef, ei)
        assert ef <= ENC_HIP_VS_ATEN_TOL and ei <= ENC_HIP_VS_ATEN_TOL, (ef, ei)
    print("HIP fp32 encoder vs ATen, worst max-abs / max:", worst)


@torch.no_grad()
def test_multiscale_towers_two_source_input_equals_the_concatenation():
    """fp16 MultiScale towers: the channel concatenations of reference extractor.py:300, 306 as two-source inputs of the
    LDS-tiled conv kernel (ramp_conv_job.x2) against torch.cat copies fed to the same kernel: bit-equal maps"""
    from rampvo_amd import conv_hip
    from rampvo_amd.synthetic import SyntheticStream, make_network
    stream = SyntheticStream(96, 128, 3, seed=6)
    outs = {}
    old = conv_hip._MS_PAIR
    try:
        for pair in (True, False):
            conv_hip._MS_PAIR = pair
            enc = make_network("MultiScale").patchify.encoder
            enc.mixed_precision = True
            res = []
            for t in range(3):
                im, ev, _, _ = stream.frame(t)
                f, i = enc(events=ev.cuda(), images=im.cuda(), mask=torch.tensor([True]), reinit_hidden=(t == 0), out_scale=0.25)
                res.append((f.clone(), i.clone()))
            outs[pair] = res
    finally:
        conv_hip._MS_PAIR = old
    for (f0, i0), (f1, i1) in zip(outs[True], outs[False]):
        assert f0.dtype == torch.float16 and torch.equal(f0, f1) and torch.equal(i0, i1)


@pytest.mark.parametrize("mode,fp8,hw", [("SingleScale", False, (96, 128)), ("MultiScale", False, (96, 128)),
                                         ("SingleScale", True, (96, 128)), ("SingleScale", False, (104, 136))])
def test_residual_block_tail_in_the_next_layers_load_equals_the_launch_of_its_own(mode, fp8, hw):
    """fp16 towers (accumulator-mode InstanceNorm): relu(skip + relu(norm2(conv2))) of reference extractor.py:49-57 formed by
    the NEXT layer's conv kernel while it stages its input tile (ramp_conv_job.skip / .mat: the stride-1 consumer also
    writes it out for the following block's skip) against ramp_norm_add_relu_f16_acc launches: bit-equal maps, over
    several frames, on the f16 and the fp8 MFMA, and at a map size with ragged tiles"""
    from rampvo_amd import conv_hip
    from rampvo_amd.synthetic import SyntheticStream, make_network
    stream = SyntheticStream(hw[0], hw[1], 3, seed=6)
    outs = {}
    old = conv_hip._TAIL_FUSE
    try:
        for fuse in (True, False):
            conv_hip._TAIL_FUSE = fuse
            enc = make_network(mode).patchify.encoder
            enc.mixed_precision = True
            enc.fp8_mfma = fp8
            res = []
            for t in range(3):
                im, ev, _, _ = stream.frame(t)
                kw = dict(mask=torch.tensor([True])) if mode == "MultiScale" else {}
                f, i = enc(events=ev.cuda(), images=im.cuda(), reinit_hidden=(t == 0), out_scale=0.25, **kw)[:2]
                res.append((f.clone(), i.clone()))
            outs[fuse] = res
    finally:
        conv_hip._TAIL_FUSE = old
    for (f0, i0), (f1, i1) in zip(outs[True], outs[False]):
        assert f0.dtype == torch.float16 and torch.equal(f0, f1) and torch.equal(i0, i1)
        assert float(f0.float().abs().max()) > 0


@pytest.mark.parametrize("mode", ["SingleScale", "MultiScale"])
def test_front_end_graph_replay_equals_eager(mode):
    """the hipGraph-captured front end (encoder + patch selection + gathers) replays to exactly what
    the eager launches produce, frame after frame (recurrent state carried through the graph)"""
    from rampvo_amd.synthetic import SyntheticStream, make_network
    stream = SyntheticStream(96, 128, 6, seed=9)
    res = {}
    for use_graph in (False, True):
        net = make_network(mode)
        net.patchify.use_graph = use_graph
        out = []
        with torch.no_grad():
            for t in range(6):
                im, ev, _, _ = stream.frame(t)
                r = net.patchify(input_=(ev.cuda(), im.cuda(), torch.tensor([True])), patches_per_image=8,
                                 event_bias=True, reinit_hidden=(t == 0))
                out.append([x.float().clone() for x in r])
        res[use_graph] = out
        if use_graph:
            assert len(net.patchify._graphs) == 1      # frames 2.. ran as replays
    for a, b in zip(res[False], res[True]):
        for x, y in zip(a, b):
            assert torch.equal(x, y)
