"""GPU parity tests of the fp32-accuracy update operator (csrc/update_x3.hip: Linear layers on the f16 matrix cores from
split fp32 operands) against fp64 / fp32 PyTorch evaluations of the reference expressions (ramp/net.py:69-90,
ramp/blocks.py:15-50).  Tolerances are stated per test."""
import ctypes

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

# one fused chain (up to 6 Linear layers, 2 LayerNorms) vs the fp64 evaluation of the same expressions, relative to the output's
# scale: measured <= 2e-6 on MI355X; an fp32 torch evaluation of the same chain sits at <= 1e-6 from fp64
X3_CHAIN_TOL = 1e-5


def _pack_ptrs(ts):
    return (ctypes.c_void_p * len(ts))(*[t.data_ptr() for t in ts])


@torch.no_grad()
def test_x3_linear_against_fp64_over_operand_ranges():
    """one Linear layer (ramp_x3_linear) on operands that stress the split: O(1) rows, rows of tiny values (below the fp16
    normal range), rows of large values (~1e3), weights with a wide dynamic range; error relative to sum |x| |w| must be
    of fp32 class (the 22-bit operands give <= 2^-21 per product; measured ~3e-7), against ~2.5e-4 for fp16 operands"""
    from rampvo_amd._lib import check, lib, ptr, stream
    from rampvo_amd.update_fused import pack_linear_x3
    g = torch.Generator().manual_seed(3)
    rows = 1000
    x = torch.randn(rows, 384, generator=g)
    x[100:200] *= 1e-6
    x[200:300] *= 1e3
    x[300:400] *= torch.exp(4 * torch.randn(100, 384, generator=g))        # wide range inside a row
    x.clamp_(-6e4, 6e4)                                                     # (the documented operand range: |x| < 65504)
    x[400] = 0
    W = torch.randn(384, 384, generator=g) * 0.05 * torch.exp(2 * torch.randn(384, 384, generator=g))
    b = torch.randn(384, generator=g)
    xc, bc = x.cuda(), b.cuda()
    pack = pack_linear_x3(W.cuda())
    y = torch.zeros(rows, 384, device="cuda")
    nrows = torch.tensor([rows - 7], dtype=torch.int32, device="cuda")
    check(lib().ramp_x3_linear(ptr(xc), ptr(pack), ptr(bc), ptr(y), rows, ptr(nrows), stream()), "ramp_x3_linear")
    assert float(y[rows - 7:].abs().max()) == 0.0                            # rows >= *rows_dev are not computed
    n = rows - 7
    ref = (x.double() @ W.double().t() + b.double())[:n]
    mag = (x.double().abs() @ W.double().abs().t() + b.double().abs())[:n]
    # an operand below the fp16 normal range (6.1e-5) is carried by the low plane alone, i.e. with 11 bits: an ABSOLUTE error
    # of <= 3e-8 per operand (fp32 class next to O(1) activations), which a row made of such operands only shows as 2^-11
    # relative -- rows 100..199 are bounded by that floor, everything else by the 22-bit products
    floor = (torch.full_like(x, 3.0e-8).double() @ W.double().abs().t())[:n]
    e = (y[:n].cpu().double() - ref).abs()
    e32 = ((x @ W.t() + b)[:n].double() - ref).abs()
    tiny = torch.zeros(n, dtype=torch.bool)
    tiny[100:200] = True
    err = (e / mag)[~tiny].max().item()
    err32 = (e32 / mag)[~tiny].max().item()
    err_tiny = (e / (floor + 1e-6 * mag))[tiny].max().item()
    print("x3 linear: max err / sum|x||w| = %.2e (torch fp32 on the CPU: %.2e); rows of sub-normal-range operands: %.2f of the "
          "3e-8-per-operand floor" % (err, err32, err_tiny))
    assert err <= 2e-6, err
    assert err_tiny <= 1.0, err_tiny


@pytest.mark.parametrize("E", [1003, 5408, 41003])
@torch.no_grad()
def test_x3_update_chains_against_fp64_torch(E):
    """every fused chain of the fp32 operator on its own, at sizes that are no multiple of the 64-row tile, against the
    fp64 evaluation of the reference module (same seeded weights)"""
    from rampvo_amd._lib import check, lib, ptr, stream
    from rampvo_amd.synthetic import make_network
    net = make_network("SingleScale")
    fu = net.update.fused(torch.float32)
    assert fu.use_x3
    w = fu.weights()
    ref = make_network("SingleScale").update.double()
    g = torch.Generator().manual_seed(23)
    G = 57
    rnd = lambda *s, sc=0.5: (torch.randn(*s, generator=g) * sc).cuda()
    worst = {}

    def cmp(name, got, exp, tol=X3_CHAIN_TOL):
        scale = float(exp.abs().max())
        err = float((got.double() - exp).abs().max()) / scale
        worst[name] = err
        assert err <= tol, (name, err, worst)

    # --- gru: LN((x + h0[gid0]) + h1[gid1]) -> GatedResidual -> LN -> GatedResidual, relu copy, heads
    x32, h0, h1 = rnd(E, 384), rnd(G, 384), rnd(G + 3, 384)
    gid0 = torch.randint(0, G, (E,), generator=g).int().cuda()
    gid1 = torch.randint(0, G + 3, (E,), generator=g).int().cuda()
    out32 = torch.empty(E, 384, device="cuda")
    relu32 = torch.empty(E, 384, device="cuda")
    _, _, wptr, bptr = w["gru_pack"]
    ln1, ln2 = w["ln1"], w["ln2"]
    check(lib().ramp_x3_gru(ptr(x32), ptr(h0), ptr(gid0), ptr(h1), ptr(gid1), ptr(ln1[0]), ptr(ln1[1]), float(ln1[2]), wptr,
                            bptr, ptr(ln2[0]), ptr(ln2[1]), float(ln2[2]), ptr(out32), ptr(relu32), E, None, None, None, None,
                            None, 3, 0.0, 0.0, stream()), "ramp_x3_gru")
    exp = ref.gru((x32.double() + h0.double()[gid0.long()]) + h1.double()[gid1.long()])
    cmp("gru", out32, exp)
    cmp("gru_relu", relu32, torch.relu(exp))
    hw, hb = w["heads_pack"]
    coords = (torch.rand(E, 2, 3, 3, generator=g) * 60).cuda()
    target, weight = torch.empty(1, E, 2, device="cuda"), torch.empty(1, E, 2, device="cuda")
    out_h = torch.empty(E, 384, device="cuda")
    check(lib().ramp_x3_gru(ptr(x32), ptr(h0), ptr(gid0), ptr(h1), ptr(gid1), ptr(ln1[0]), ptr(ln1[1]), float(ln1[2]), wptr,
                            bptr, ptr(ln2[0]), ptr(ln2[1]), float(ln2[2]), ptr(out_h), None, E, ptr(hw), ptr(hb), ptr(coords),
                            ptr(target), ptr(weight), 3, 40.0, 30.0, stream()), "ramp_x3_gru")
    assert torch.equal(out_h, out32)
    t_exp = coords[:, :, 1, 1].double() + ref.d[1](torch.relu(exp))
    w_exp = torch.sigmoid(ref.w[1](torch.relu(exp)))
    inside = (t_exp[:, 0] >= 0) & (t_exp[:, 1] >= 0) & (t_exp[:, 0] <= 40.0) & (t_exp[:, 1] <= 30.0)     # utils.py:557-570
    cmp("heads_target", target[0], t_exp, tol=2e-6)
    far = ((t_exp - torch.tensor([40.0, 30.0], device="cuda")).abs().min(-1).values > 1e-3) & (t_exp.abs().min(-1).values > 1e-3)
    assert float(((weight[0].double() - w_exp * inside[:, None].double()).abs().max(-1).values * far.double()).max()) <= 2e-6
    # without the prologue: x32 is already gru[0]'s output
    xin = ref.gru[0](x32.double()).float()
    check(lib().ramp_x3_gru(ptr(xin), None, None, None, None, None, None, 0.0, wptr, bptr, ptr(ln2[0]), ptr(ln2[1]),
                            float(ln2[2]), ptr(out32), ptr(relu32), E, None, None, None, None, None, 3, 0.0, 0.0, stream()),
          "ramp_x3_gru")
    cmp("gru_noprologue", out32, ref.gru[3](ref.gru[2](ref.gru[1](xin.double()))))

    # --- c1 / c2: net + Lb(relu(La(mask * net[idx])))
    net_in = rnd(E, 384)
    idx = torch.randint(-1, E, (E,), generator=g).cuda()
    idx[::5] = -1
    for name, seq in (("c1_pack", ref.c1), ("c2_pack", ref.c2)):
        wa, ba, wb, bb = w[name]
        o = torch.empty_like(net_in)
        check(lib().ramp_x3_nbr(ptr(net_in), ptr(idx), ptr(wa), ptr(ba), ptr(wb), ptr(bb), ptr(o), E, stream()), "ramp_x3_nbr")
        gathered = net_in.double()[idx.clamp(min=0)] * (idx >= 0).double()[:, None]
        cmp("nbr_" + name, o, net_in.double() + seq(gathered))

    # --- correlation MLP + Update.norm: LN((net[map] + inp[idx % mod]) + corr-MLP(corr))
    corr = torch.nn.functional.pad(rnd(E, 882, sc=2.0), (0, 14)).contiguous()
    state = rnd(700, 384)
    net_map = torch.randint(-1, 700, (E,), generator=g).cuda()
    net_map[-3:] = 699
    table = rnd(300, 384)
    inp_idx = torch.randint(0, 5000, (E,), generator=g).cuda()
    w1, b1 = w["corr1_pack"]
    w2, b2, w3, b3 = w["tail_pack"]
    ln, nm = w["corr_ln"], w["norm"]
    o = torch.empty(E, 384, device="cuda")
    check(lib().ramp_x3_corr_mlp(ptr(corr), 896, ptr(w1), ptr(b1), ptr(w2), ptr(b2), ptr(w3), ptr(b3), ptr(ln[0]), ptr(ln[1]),
                                 float(ln[2]), ptr(state), ptr(net_map), ptr(table), ptr(inp_idx), 300, ptr(nm[0]), ptr(nm[1]),
                                 float(nm[2]), ptr(o), E, stream()), "ramp_x3_corr_mlp")
    st = state.double()[net_map.clamp(min=0)] * (net_map >= 0).double()[:, None]
    cmp("corr_mlp", o, ref.norm(st + table.double()[inp_idx % 300] + ref.corr(corr[:, :882].double())))
    # zero state / identity context (the motion probe's form)
    tab2 = rnd(E, 384)
    check(lib().ramp_x3_corr_mlp(ptr(corr), 896, ptr(w1), ptr(b1), ptr(w2), ptr(b2), ptr(w3), ptr(b3), ptr(ln[0]), ptr(ln[1]),
                                 float(ln[2]), None, None, ptr(tab2), None, 0, ptr(nm[0]), ptr(nm[1]), float(nm[2]), ptr(o), E,
                                 stream()), "ramp_x3_corr_mlp")
    cmp("corr_mlp_zero_state", o, ref.norm(tab2.double() + ref.corr(corr[:, :882].double())))

    # --- SoftAgg: [f | g] rows (+ the expand-and-add of the previous one), segment softmax, h
    order = torch.argsort(gid0.long(), stable=True).int()
    counts = torch.bincount(gid0.long(), minlength=G)
    seg = torch.zeros(G + 1, dtype=torch.int32, device="cuda")
    seg[1:] = torch.cumsum(counts, 0).int()
    ngroups = torch.tensor([G], dtype=torch.int32, device="cuda")
    for name, hname, agg, add in (("kk_fg_pack", "kk_h_pack", ref.agg_kk, False), ("ij_fg_pack", "ij_h_pack", ref.agg_ij, True)):
        wf, bf, wg, bg = w[name]
        xs = x32.clone()
        fg = torch.empty(E, 768, device="cuda")
        check(lib().ramp_x3_fg(ptr(xs), ptr(h1) if add else None, ptr(gid1) if add else None, ptr(xs) if add else None, ptr(wf),
                               ptr(bf), ptr(wg), ptr(bg), ptr(fg), E, stream()), "ramp_x3_fg")
        xe = x32.double() + (h1.double()[gid1.long()] if add else 0)
        if add:
            cmp("fg_x_" + name, xs, xe, tol=1e-7)
        cmp("fg_" + name, fg, torch.cat([agg.f(xe), agg.g(xe)], 1))
        y = torch.empty(G + 4, 384, device="cuda")
        check(lib().ramp_x3_segment_softmax(ptr(fg), ptr(order), ptr(seg), ptr(ngroups), ptr(y), G + 4, stream()), "seg")
        assert float(y[G:].abs().max()) == 0.0
        fgd = fg.double()
        ye = torch.zeros(G, 384, dtype=torch.float64, device="cuda")
        for k in range(G):
            m = gid0 == k
            if m.any():
                ye[k] = (torch.softmax(fgd[m, 384:], 0) * fgd[m, :384]).sum(0)
        cmp("seg_" + name, y[:G], ye, tol=2e-6)
        hy = torch.zeros(G + 4, 384, device="cuda")
        hp = w[hname]
        check(lib().ramp_x3_linear(ptr(y), ptr(hp[0]), ptr(hp[1]), ptr(hy), G + 4, ptr(ngroups), stream()), "ramp_x3_linear")
        cmp("h_" + name, hy[:G], agg.h(y[:G].double()))
    print("x3 update chains vs fp64 torch, max err / output scale:", {k: float("%.2e" % v) for k, v in worst.items()})


@torch.no_grad()
def test_x3_operator_equals_the_library_gemm_operator():
    """Update.forward in fp32 through the x3 chains (the default) and through library GEMMs + row kernels (RAMP_X3=0's
    path, here by flipping the instance's switch): the same operator to fp32-GEMM accuracy on a structured graph"""
    from pipeline_checks import gold
    from rampvo_amd.net import GraphPlan
    from rampvo_amd.synthetic import make_network
    net = make_network("SingleScale")
    g = torch.Generator().manual_seed(5)
    gd = gold("update_op.npz")                     # the structured graph of the reference-generated operator fixture
    ii, jj, kk = gd["ii"], gd["jj"], gd["kk"]
    E = len(ii)
    cu = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    plan = GraphPlan.build(cu(ii), cu(jj), cu(kk))
    h = (torch.randn(1, E, 384, generator=g) * 0.5).cuda()
    inp = (torch.randn(1, E, 384, generator=g) * 0.5).cuda()
    corr = (torch.randn(1, E, 882, generator=g) * 2.0).cuda()
    fu = net.update.fused(torch.float32)
    outs = []
    for x3 in (True, False):
        fu.use_x3, fu._w = x3, None
        o, (d, wgt, _) = net.update(h, inp, corr, None, cu(ii), cu(jj), cu(kk), plan=plan)
        outs.append((o.clone(), d.clone(), wgt.clone()))
    fu.use_x3, fu._w = True, None
    for a, b, tol in zip(outs[0], outs[1], (2e-5, 2e-5, 2e-5)):
        err = float((a - b).abs().max() / b.abs().max())
        assert err <= tol, err
