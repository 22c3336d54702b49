"""Fused forward of the update operator (reference: ramp/net.py:69-90).

~19 GEMMs (hipBLASLt through torch, bias / bias+ReLU epilogues) stitched by the row-fused HIP
kernels of csrc/update.hip: every gather, residual add, LayerNorm, gate and dtype cast between
two GEMMs is ONE kernel.  The hidden state stays fp32; GEMM inputs/outputs are ``dtype`` (half
under MIXED_PRECISION -- what the reference's autocast does -- or float).  Weight copies in
``dtype`` (f|g and the two heads stacked) are cached per module.
"""
import os

import torch
import torch.nn.functional as F

from . import _lib
from ._lib import check, lib, ptr, stream

_code = {torch.float32: _lib.RAMP_F32, torch.float16: _lib.RAMP_F16}


CORR_ROW = 896     # correlation rows are padded from 882 to 896 elements on the GPU (zero tail)


def pack_linear_f16(weight):
    """nn.Linear weight [N, K] -> fp16 MFMA B fragments [K/32][N/16][64 lanes][8]:
    lane (q = lane >> 4, j = lane & 15) of fragment (ks, nt) holds W[16 nt + j][32 ks + 8 q .. + 8]"""
    w = weight.detach().half()
    N, K = w.shape
    assert N % 16 == 0 and K % 32 == 0
    return w.view(N // 16, 16, K // 32, 4, 8).permute(2, 0, 3, 1, 4).contiguous()


def pack_linear_x3(weight):
    """nn.Linear weight [N = 384, K] (K a multiple of 32) -> the split-fp16 pack of csrc/update_x3.hip: a flat fp16 tensor
    [K/32][N/16][2][64 lanes][8] -- plane 0 = fp16(W 2^s), plane 1 = fp16(W 2^s - plane 0), fragment order as
    pack_linear_f16 -- followed by the float 2^-s (as two fp16 slots) and padding to 16 bytes; s is the power of two that
    puts max |W| into [2^12, 2^13): the high part, the high part times 2^-11 and the low part are then fp16 NORMAL numbers"""
    import math
    w = weight.detach().float()
    N, K = w.shape
    assert N == 384 and K % 32 == 0
    amax = float(w.abs().max())
    s = 13 - math.frexp(amax)[1] if amax > 0 and math.isfinite(amax) else 0      # amax = m 2^e, m in [0.5, 1)
    ws = w * (2.0 ** s)                                    # exact
    hi = ws.half()
    lo = (ws - hi.float()).half()
    frag = lambda t: t.view(N // 16, 16, K // 32, 4, 8).permute(2, 0, 3, 1, 4)          # [K/32][N/16][4][16][8]
    planes = torch.stack([frag(hi), frag(lo)], dim=2).contiguous().view(-1)                 # [K/32][N/16][2][64][8]
    inv = torch.tensor([2.0 ** -s], dtype=torch.float32, device=w.device).view(torch.float16)
    return torch.cat([planes, inv, torch.zeros(6, dtype=torch.float16, device=w.device)]).contiguous()


class FusedUpdate:
    def __init__(self, update, dtype):
        self.m = update
        self.dtype = dtype
        self._w = None
        self._key = None
        self._params = list(update.parameters())     # module structure is fixed; values are tracked by key
        self._act_ok = True
        self.use_mlp = True          # fp16: the fused GEMM-chain kernels of csrc/update_mlp.hip (instance switches, for tests:
        self.use_corr_mlp = True     # False falls back to library GEMMs + the row kernels of csrc/update.hip stage by stage)
        self.use_softagg = True      # SoftAgg without the [f | g] rows
        # fp32: the Linear layers on the f16 matrix cores from split operands (csrc/update_x3.hip), fused chains as on the
        # fp16 path; RAMP_X3=0: library GEMMs + the row kernels of csrc/update.hip (A/B runs)
        self.use_x3 = dtype == torch.float32 and os.environ.get("RAMP_X3", "1") == "1"
        self.before_gru = None                   # optional callable run right before a stage is enqueued
        self.hook_at = "gru"

    # ------------------------------------------------------------------ weights
    def weights(self):
        key = tuple([(p.data_ptr(), p._version) for p in self._params])
        if self._w is not None and key == self._key:
            return self._w
        m, T = self.m, self.dtype
        c = lambda t: t.detach().to(T).contiguous()
        f = lambda t: t.detach().float().contiguous()
        w = dict(
            corr0=(c(m.corr[0].weight), c(m.corr[0].bias)),
            corr0_pad=(c(F.pad(m.corr[0].weight, (0, CORR_ROW - m.corr[0].weight.shape[1]))), c(m.corr[0].bias)),
            corr2=(c(m.corr[2].weight), c(m.corr[2].bias)),
            corr_ln=(f(m.corr[3].weight), f(m.corr[3].bias), m.corr[3].eps),
            corr5=(c(m.corr[5].weight), c(m.corr[5].bias)),
            norm=(f(m.norm.weight), f(m.norm.bias), m.norm.eps),
            c1a=(c(m.c1[0].weight), c(m.c1[0].bias)), c1b=(c(m.c1[2].weight), c(m.c1[2].bias)),
            c2a=(c(m.c2[0].weight), c(m.c2[0].bias)), c2b=(c(m.c2[2].weight), c(m.c2[2].bias)),
            kk_fg=(c(torch.cat([m.agg_kk.f.weight, m.agg_kk.g.weight], 0)),
                   c(torch.cat([m.agg_kk.f.bias, m.agg_kk.g.bias], 0))),
            kk_h=(c(m.agg_kk.h.weight), c(m.agg_kk.h.bias)),
            ij_fg=(c(torch.cat([m.agg_ij.f.weight, m.agg_ij.g.weight], 0)),
                   c(torch.cat([m.agg_ij.f.bias, m.agg_ij.g.bias], 0))),
            ij_h=(c(m.agg_ij.h.weight), c(m.agg_ij.h.bias)),
            ln1=(f(m.gru[0].weight), f(m.gru[0].bias), m.gru[0].eps),
            g1_gate=(c(m.gru[1].gate[0].weight), c(m.gru[1].gate[0].bias)),
            g1_r1=(c(m.gru[1].res[0].weight), c(m.gru[1].res[0].bias)),
            g1_r2=(c(m.gru[1].res[2].weight), c(m.gru[1].res[2].bias)),
            ln2=(f(m.gru[2].weight), f(m.gru[2].bias), m.gru[2].eps),
            g2_gate=(c(m.gru[3].gate[0].weight), c(m.gru[3].gate[0].bias)),
            g2_r1=(c(m.gru[3].res[0].weight), c(m.gru[3].res[0].bias)),
            g2_r2=(c(m.gru[3].res[2].weight), c(m.gru[3].res[2].bias)),
            heads=(c(torch.cat([m.d[1].weight, m.w[1].weight], 0)), c(torch.cat([m.d[1].bias, m.w[1].bias], 0))),
        )
        if T == torch.float16:
            # fused GEMM chains (csrc/update_mlp.hip): weights in MFMA fragment order, biases as the fp32
            # values of their fp16 roundings (they are half tensors under the reference's autocast)
            import ctypes
            gru = [m.gru[1].gate[0], m.gru[1].res[0], m.gru[1].res[2], m.gru[3].gate[0], m.gru[3].res[0], m.gru[3].res[2]]
            wp = [pack_linear_f16(l.weight) for l in gru]
            bs = [l.bias.detach().half().float().contiguous() for l in gru]
            w["gru_pack"] = (wp, bs, (ctypes.c_void_p * 6)(*[t.data_ptr() for t in wp]),
                             (ctypes.c_void_p * 6)(*[t.data_ptr() for t in bs]))
            hb = lambda l: l.bias.detach().half().float().contiguous()
            w["c1_pack"] = (pack_linear_f16(m.c1[0].weight), hb(m.c1[0]), pack_linear_f16(m.c1[2].weight), hb(m.c1[2]))
            w["c2_pack"] = (pack_linear_f16(m.c2[0].weight), hb(m.c2[0]), pack_linear_f16(m.c2[2].weight), hb(m.c2[2]))
            for name, agg in (("kk_fg_pack", m.agg_kk), ("ij_fg_pack", m.agg_ij)):
                w[name] = (pack_linear_f16(agg.f.weight), hb(agg.f), pack_linear_f16(agg.g.weight), hb(agg.g))
            w["kk_h_pack"] = (pack_linear_f16(m.agg_kk.h.weight), hb(m.agg_kk.h))
            w["ij_h_pack"] = (pack_linear_f16(m.agg_ij.h.weight), hb(m.agg_ij.h))
            w["corr1_pack"] = (pack_linear_f16(F.pad(m.corr[0].weight, (0, CORR_ROW - m.corr[0].weight.shape[1]))),
                               hb(m.corr[0]))
            w["heads_pack"] = (w["heads"][0], w["heads"][1].float().contiguous())
            w["tail_pack"] = (pack_linear_f16(m.corr[2].weight), hb(m.corr[2]), pack_linear_f16(m.corr[5].weight),
                              hb(m.corr[5]))
        if self.use_x3:
            import ctypes
            f32 = lambda t: t.detach().float().contiguous()
            px = lambda l, pad=0: (pack_linear_x3(F.pad(l.weight, (0, pad)) if pad else l.weight), f32(l.bias))
            gru = [m.gru[1].gate[0], m.gru[1].res[0], m.gru[1].res[2], m.gru[3].gate[0], m.gru[3].res[0], m.gru[3].res[2]]
            wp = [pack_linear_x3(l.weight) for l in gru]
            bs = [f32(l.bias) for l in gru]
            w["gru_pack"] = (wp, bs, (ctypes.c_void_p * 6)(*[t.data_ptr() for t in wp]),
                             (ctypes.c_void_p * 6)(*[t.data_ptr() for t in bs]))
            w["c1_pack"] = px(m.c1[0]) + px(m.c1[2])
            w["c2_pack"] = px(m.c2[0]) + px(m.c2[2])
            w["kk_fg_pack"] = px(m.agg_kk.f) + px(m.agg_kk.g)
            w["ij_fg_pack"] = px(m.agg_ij.f) + px(m.agg_ij.g)
            w["kk_h_pack"], w["ij_h_pack"] = px(m.agg_kk.h), px(m.agg_ij.h)
            w["corr1_pack"] = px(m.corr[0], CORR_ROW - m.corr[0].weight.shape[1])
            w["tail_pack"] = px(m.corr[2]) + px(m.corr[5])
            w["heads_pack"] = (f32(torch.cat([m.d[1].weight, m.w[1].weight], 0)), f32(torch.cat([m.d[1].bias, m.w[1].bias], 0)))
        self._w, self._key = w, key
        return w

    # --------------------------------------------------------------- primitives
    def lin(self, x, wb):
        return F.linear(x, wb[0], wb[1])

    def lin_relu(self, x, wb):
        if self._act_ok:
            try:
                return torch._addmm_activation(wb[1], x, wb[0].t(), use_gelu=False)
            except Exception:
                self._act_ok = False
        return F.relu_(F.linear(x, wb[0], wb[1]))

    def row_fuse(self, E, A=None, B=None, C=None, idxB=None, idxB32=None, modB=0, idxC32=None, ln=None, relu=False,
                 want_f32=False, want_t=False, out_f32=None, idxA=None):
        dev = (A if A is not None else B).device
        if want_f32 and out_f32 is None:
            out_f32 = torch.empty(E, 384, dtype=torch.float32, device=dev)
        out_t = torch.empty(E, 384, dtype=self.dtype, device=dev) if want_t else None
        check(lib().ramp_upd_row_fuse(ptr(A), ptr(idxA), ptr(B), ptr(C), ptr(idxB), ptr(idxB32), int(modB), None,
                                      ptr(idxC32),
                                      ptr(ln[0]) if ln else None, ptr(ln[1]) if ln else None,
                                      float(ln[2]) if ln else 0.0, int(relu), ptr(out_f32), ptr(out_t), E,
                                      _code[self.dtype], stream()), "ramp_upd_row_fuse")
        return out_f32, out_t

    def gather_mask(self, X, idx, E):
        out = torch.empty(E, 384, dtype=self.dtype, device=X.device)
        check(lib().ramp_upd_gather_mask(ptr(X), ptr(idx), ptr(out), E, _code[self.dtype], stream()),
              "ramp_upd_gather_mask")
        return out

    def gated(self, X, G, R, E, ln=None, want_f32=True, want_t=False, want_relu=False, out=None):
        dev = X.device
        o32 = (out if out is not None else torch.empty(E, 384, dtype=torch.float32, device=dev)) if want_f32 else None
        ot = torch.empty(E, 384, dtype=self.dtype, device=dev) if want_t else None
        orl = torch.empty(E, 384, dtype=self.dtype, device=dev) if want_relu else None
        check(lib().ramp_upd_gated(ptr(X), ptr(G), ptr(R), ptr(ln[0]) if ln else None, ptr(ln[1]) if ln else None,
                                   float(ln[2]) if ln else 0.0, ptr(o32), ptr(ot), ptr(orl), E, _code[self.dtype],
                                   stream()), "ramp_upd_gated")
        return o32, ot, orl

    def fg(self, net32, add_t, add_idx, pack, E):
        """[f(x) | g(x)] of x = net32 (+ add_t[add_idx], written back to net32): csrc/update_mlp.hip::upd_fg_kernel"""
        out = torch.empty(E, 768, dtype=self.dtype, device=net32.device)
        wf, bf, wg, bg = pack
        check(lib().ramp_upd_fg(ptr(net32), ptr(add_t), ptr(add_idx), ptr(net32) if add_t is not None else None,
                                ptr(wf), ptr(bf), ptr(wg), ptr(bg), ptr(out), E, stream()), "ramp_upd_fg")
        return out

    def seg(self, fg, groups, max_groups):
        y = torch.empty(max(max_groups, 1), 384, dtype=self.dtype, device=fg.device)     # the kernel writes every row
        check(lib().ramp_upd_segment_softmax(ptr(fg), ptr(groups.order), ptr(groups.seg_start), ptr(groups.ngroups),
                                             ptr(y), int(max_groups), _code[self.dtype], stream()),
              "ramp_upd_segment_softmax")
        return y

    def softagg(self, net32, add_t, add_idx, fg_pack, h_pack, groups, max_groups, E):
        """h(segment softmax-sum of f(x), g(x)) for x = net32 (+ add_t[add_idx]) without the [E, 768] rows:
        csrc/update_mlp.hip::upd_softagg_kernel + upd_softagg_finish_kernel (the device-resident step runs the same two)"""
        G = max(int(max_groups), 1)
        rows = lib().ramp_upd_softagg_frag_rows(E, G)
        frag = torch.empty(rows, 3, 384, dtype=torch.float32, device=net32.device)
        wf, bf, wg, bg = fg_pack
        check(lib().ramp_upd_softagg(ptr(net32), ptr(add_t), ptr(add_idx), ptr(groups.order), ptr(groups.gid), ptr(wf), ptr(bf),
                                     ptr(wg), ptr(bg), ptr(frag), E, stream()), "ramp_upd_softagg")
        hy = torch.empty(G, 384, dtype=self.dtype, device=net32.device)
        check(lib().ramp_upd_softagg_finish(ptr(frag), ptr(groups.seg_start), ptr(groups.ngroups), ptr(h_pack[0]),
                                            ptr(h_pack[1]), ptr(hy), G, stream()), "ramp_upd_softagg_finish")
        return hy

    def h_lin(self, y, pack, groups):
        """SoftAgg's `h` Linear on the group table, rows below the device-side group count only
        (csrc/update_mlp.hip::upd_linear_kernel; the device-resident step runs the same kernel)"""
        out = torch.empty_like(y)
        check(lib().ramp_upd_linear(ptr(y), ptr(pack[0]), ptr(pack[1]), ptr(out), y.shape[0], ptr(groups.ngroups),
                                    stream()), "ramp_upd_linear")
        return out

    # ------------------------------------------------------------------ forward
    def hidden(self, net, inp_table, inp_idx, inp_mod, corr, plan, net_map=None, heads_at=None, net32_buf=None, out32_buf=None):
        """net [*,384] fp32 or None (zeros), row net_map[e] of it per edge when net_map is given (-1: zero
        row); inp = inp_table[inp_idx % inp_mod] (or inp_table rows when inp_idx is None); corr [E,882] in
        self.dtype.  Returns (net_out fp32 [E,384], relu copy T).  heads_at = (coords [E,2,P,P], wd, ht): the fused gru
        launch also forms the heads and target / weight (left in self.last_tw; the relu copy is then None)."""
        self.last_tw, self._heads_at = None, heads_at
        # (net32_buf / out32_buf: capacity-sized [>= E, 384] fp32 buffers for the state after the first LayerNorm and for the
        # result -- the device-resident fp32 step gathers through index rows beyond the live factor count, which must stay
        # inside an allocation; the non-fused path only)
        self._bufs = (net32_buf, out32_buf)
        w = self.weights()
        E = corr.shape[0]
        if self.use_x3:
            return self._hidden_x3(w, net, inp_table, inp_idx, inp_mod, corr, plan, net_map, heads_at)
        if "tail_pack" in w and self.use_mlp and corr.shape[1] == CORR_ROW and self.use_corr_mlp:
            # the whole correlation MLP (3 Linear, LayerNorm, ReLUs) + net + inp + c + LayerNorm: one launch
            w1, b1 = w["corr1_pack"]
            w2, b2, w3, b3 = w["tail_pack"]
            ln, nm = w["corr_ln"], w["norm"]
            net32 = torch.empty(E, 384, dtype=torch.float32, device=corr.device)
            check(lib().ramp_upd_corr_mlp(ptr(corr), CORR_ROW, ptr(w1), ptr(b1), ptr(w2), ptr(b2), ptr(w3), ptr(b3),
                                          ptr(ln[0]), ptr(ln[1]), float(ln[2]), ptr(net), ptr(net_map), ptr(inp_table),
                                          ptr(inp_idx), int(inp_mod or 0), ptr(nm[0]), ptr(nm[1]), float(nm[2]),
                                          ptr(net32), E, stream()), "ramp_upd_corr_mlp")
            c = None
        else:
            c = self.lin_relu(corr, w["corr0_pad"] if corr.shape[1] == CORR_ROW else w["corr0"])
        if c is None:
            pass
        else:
            c = self.lin(c, w["corr2"])
            _, c = self.row_fuse(E, B=c, ln=w["corr_ln"], relu=True, want_t=True)
            c = self.lin(c, w["corr5"])
            net32, _ = self.row_fuse(E, A=net, idxA=net_map, B=inp_table, idxB=inp_idx, modB=inp_mod, C=c,
                                     ln=w["norm"], want_f32=True, out_f32=net32_buf[:E] if net32_buf is not None else None)
        # temporal neighbours (net.py:77-82); plan.ix_raw / jx_raw keep the -1 markers
        if "c1_pack" in w and self.use_mlp:
            # gather + 2 Linear + residual add per launch; ping-pong between two state buffers
            tmp = torch.empty_like(net32)
            net_t = torch.empty(E, 384, dtype=self.dtype, device=net32.device)
            if self.before_gru is not None and self.hook_at == "nbr":
                self.before_gru()
            wa, ba, wb, bb = w["c1_pack"]
            wa2, ba2, wb2, bb2 = w["c2_pack"]
            check(lib().ramp_upd_nbr(ptr(net32), ptr(plan.ix_raw), ptr(wa), ptr(ba), ptr(wb), ptr(bb), ptr(tmp), None,
                                     E, stream()), "ramp_upd_nbr")
            check(lib().ramp_upd_nbr(ptr(tmp), ptr(plan.jx_raw), ptr(wa2), ptr(ba2), ptr(wb2), ptr(bb2), ptr(net32),
                                     None, E, stream()), "ramp_upd_nbr")
            return self._tail(w, E, net32, None, plan)
        g = self.gather_mask(net32, plan.ix_raw, E)
        y = self.lin(self.lin_relu(g, w["c1a"]), w["c1b"])
        self.row_fuse(E, A=net32, B=y, out_f32=net32)
        g = self.gather_mask(net32, plan.jx_raw, E)
        y = self.lin(self.lin_relu(g, w["c2a"]), w["c2b"])
        _, net_t = self.row_fuse(E, A=net32, B=y, out_f32=net32, want_t=True)
        return self._tail(w, E, net32, net_t, plan)

    def _hidden_x3(self, w, net, inp_table, inp_idx, inp_mod, corr, plan, net_map, heads_at):
        """the fp32 operator as six fused launches + the two SoftAgg tables (csrc/update_x3.hip); the device-resident step
        (csrc/track.hip) makes the same launches"""
        E, dev = corr.shape[0], corr.device
        f32 = torch.float32
        if E == 0:
            z = torch.zeros(0, 384, dtype=f32, device=dev)
            self.last_tw = (torch.zeros(1, 0, 2, dtype=f32, device=dev), torch.zeros(1, 0, 2, dtype=f32, device=dev))
            return z, None
        if corr.shape[1] != CORR_ROW:          # (a caller with dense [E, 882] rows: zero tail for the 16-byte row loads)
            corr = F.pad(corr, (0, CORR_ROW - corr.shape[1]))
        corr = corr.contiguous()
        inp_table = inp_table.float() if inp_table.dtype != f32 else inp_table
        w1, b1 = w["corr1_pack"]
        w2, b2, w3, b3 = w["tail_pack"]
        ln, nm = w["corr_ln"], w["norm"]
        a = torch.empty(E, 384, dtype=f32, device=dev)
        b = torch.empty(E, 384, dtype=f32, device=dev)
        check(lib().ramp_x3_corr_mlp(ptr(corr), CORR_ROW, ptr(w1), ptr(b1), ptr(w2), ptr(b2), ptr(w3), ptr(b3), ptr(ln[0]),
                                     ptr(ln[1]), float(ln[2]), ptr(net), ptr(net_map), ptr(inp_table), ptr(inp_idx),
                                     int(inp_mod or 0), ptr(nm[0]), ptr(nm[1]), float(nm[2]), ptr(a), E, stream()),
              "ramp_x3_corr_mlp")
        if self.before_gru is not None and self.hook_at == "nbr":
            self.before_gru()
        wa, ba, wb, bb = w["c1_pack"]
        check(lib().ramp_x3_nbr(ptr(a), ptr(plan.ix_raw), ptr(wa), ptr(ba), ptr(wb), ptr(bb), ptr(b), E, stream()), "ramp_x3_nbr")
        wa, ba, wb, bb = w["c2_pack"]
        check(lib().ramp_x3_nbr(ptr(b), ptr(plan.jx_raw), ptr(wa), ptr(ba), ptr(wb), ptr(bb), ptr(a), E, stream()), "ramp_x3_nbr")
        if self.before_gru is not None and self.hook_at == "softagg":
            self.before_gru()
        fg = torch.empty(E, 768, dtype=f32, device=dev)

        def softagg(add_t, add_idx, fg_pack, h_pack, groups, max_groups):
            G = max(int(max_groups), 1)
            wf, bf, wg, bg = fg_pack
            check(lib().ramp_x3_fg(ptr(a), ptr(add_t), ptr(add_idx), None, ptr(wf), ptr(bf), ptr(wg), ptr(bg), ptr(fg), E,
                                   stream()), "ramp_x3_fg")
            y = torch.empty(G, 384, dtype=f32, device=dev)
            check(lib().ramp_x3_segment_softmax(ptr(fg), ptr(groups.order), ptr(groups.seg_start), ptr(groups.ngroups), ptr(y),
                                                G, stream()), "ramp_x3_segment_softmax")
            hy = torch.zeros(G, 384, dtype=f32, device=dev)
            check(lib().ramp_x3_linear(ptr(y), ptr(h_pack[0]), ptr(h_pack[1]), ptr(hy), G, ptr(groups.ngroups), stream()),
                  "ramp_x3_linear")
            return hy
        # (the pair SoftAgg reads net + hkk[patch group] without writing the sum back; the gru launch forms
        # (net + hkk[.]) + hij[.] itself: the same fp32 additions in the same order)
        hkk = softagg(None, None, w["kk_fg_pack"], w["kk_h_pack"], plan.g_kk, plan.max_kk)
        hij = softagg(hkk, plan.g_kk.gid, w["ij_fg_pack"], w["ij_h_pack"], plan.g_ij, plan.max_ij)
        if self.before_gru is not None and self.hook_at == "gru":
            self.before_gru()
        _, _, wptr, bptr = w["gru_pack"]
        ln1, ln2 = w["ln1"], w["ln2"]
        out32 = b
        target = weight = relu32 = None
        hw = hb = coords = None
        wd = ht = 0.0
        P = 3
        if heads_at is not None:
            coords, wd, ht = heads_at
            P = coords.shape[-1]
            hw, hb = w["heads_pack"]
            target = torch.empty(1, E, 2, dtype=f32, device=dev)
            weight = torch.empty(1, E, 2, dtype=f32, device=dev)
        else:
            relu32 = torch.empty(E, 384, dtype=f32, device=dev)
        check(lib().ramp_x3_gru(ptr(a), ptr(hkk), ptr(plan.g_kk.gid), ptr(hij), ptr(plan.g_ij.gid), ptr(ln1[0]), ptr(ln1[1]),
                                float(ln1[2]), wptr, bptr, ptr(ln2[0]), ptr(ln2[1]), float(ln2[2]), ptr(out32), ptr(relu32), E,
                                ptr(hw), ptr(hb), ptr(coords), ptr(target), ptr(weight), int(P), float(wd), float(ht), stream()),
              "ramp_x3_gru")
        if heads_at is not None:
            self.last_tw = (target, weight)
            return out32, None
        return out32, relu32

    def _tail(self, w, E, net32, net_t, plan):
        """SoftAgg x2 and the gru block, from the state after c1 / c2"""
        # SoftAgg over patches, then over (i, j) pairs (net.py:84-85)
        if self.before_gru is not None and self.hook_at == "softagg":
            self.before_gru()
        if net_t is None and self.use_softagg and E > 0:
            hy0 = self.softagg(net32, None, None, w["kk_fg_pack"], w["kk_h_pack"], plan.g_kk, plan.max_kk, E)
            # (the device-resident step adds hy0[.] inside the next two launches instead of writing the sum back: the
            # same fp32 additions in the same order)
            self.row_fuse(E, A=net32, B=hy0, idxB32=plan.g_kk.gid, out_f32=net32)
            hy = self.softagg(net32, None, None, w["ij_fg_pack"], w["ij_h_pack"], plan.g_ij, plan.max_ij, E)
        elif net_t is None:
            # fused path: the [f | g] GEMM forms its own fp16 input tile from the fp32 state (and applies the
            # previous SoftAgg's expand-and-add on the way): no fp16 state copy, no separate row pass
            hy = self.h_lin(self.seg(self.fg(net32, None, None, w["kk_fg_pack"], E), plan.g_kk, plan.max_kk),
                            w["kk_h_pack"], plan.g_kk)
            hy = self.h_lin(self.seg(self.fg(net32, hy, plan.g_kk.gid, w["ij_fg_pack"], E), plan.g_ij, plan.max_ij),
                            w["ij_h_pack"], plan.g_ij)
        else:
            hy = self.lin(self.seg(self.lin(net_t, w["kk_fg"]), plan.g_kk, plan.max_kk), w["kk_h"])
            _, net_t = self.row_fuse(E, A=net32, B=hy, idxB32=plan.g_kk.gid, out_f32=net32, want_t=True)
            hy = self.lin(self.seg(self.lin(net_t, w["ij_fg"]), plan.g_ij, plan.max_ij), w["ij_h"])
        # gru = LN, GatedResidual, LN, GatedResidual (net.py:49-54)
        if "gru_pack" in w and self.use_mlp:
            # the chain kernel forms LN(net + hy[gid]) itself while it stages its tile: no separate row pass
            _, _, wptr, bptr = w["gru_pack"]
            out32 = torch.empty(E, 384, dtype=torch.float32, device=net32.device)
            relu_t = torch.empty(E, 384, dtype=self.dtype, device=net32.device)
            ln1 = w["ln1"]
            if self.before_gru is not None and self.hook_at == "gru":
                self.before_gru()
            heads_at = getattr(self, "_heads_at", None)
            if heads_at is not None and "heads_pack" in w and self.dtype == torch.float16:
                coords, wd, ht = heads_at
                hwt, hb = w["heads_pack"]
                target = torch.empty(1, E, 2, dtype=torch.float32, device=net32.device)
                weight = torch.empty(1, E, 2, dtype=torch.float32, device=net32.device)
                check(lib().ramp_upd_gru_heads(ptr(net32), ptr(hy), ptr(plan.g_ij.gid), ptr(ln1[0]), ptr(ln1[1]),
                                               float(ln1[2]), wptr, bptr, ptr(w["ln2"][0]), ptr(w["ln2"][1]),
                                               float(w["ln2"][2]), ptr(out32), ptr(hwt), ptr(hb), ptr(coords), ptr(target),
                                               ptr(weight), E, coords.shape[-1], float(wd), float(ht), stream()),
                      "ramp_upd_gru_heads")
                self.last_tw = (target, weight)
                return out32, None
            check(lib().ramp_upd_gru(ptr(net32), ptr(hy), ptr(plan.g_ij.gid), ptr(ln1[0]), ptr(ln1[1]), float(ln1[2]),
                                     wptr, bptr, ptr(w["ln2"][0]), ptr(w["ln2"][1]), float(w["ln2"][2]),
                                     ptr(out32), ptr(relu_t), E, stream()), "ramp_upd_gru")
            return out32, relu_t
        x32, x_t = self.row_fuse(E, A=net32, B=hy, idxB32=plan.g_ij.gid, ln=w["ln1"], out_f32=net32, want_t=True)
        gate = self.lin(x_t, w["g1_gate"])
        r = self.lin(self.lin_relu(x_t, w["g1_r1"]), w["g1_r2"])
        x32, x_t, _ = self.gated(x32, gate, r, E, ln=w["ln2"], want_t=True)
        gate = self.lin(x_t, w["g2_gate"])
        r = self.lin(self.lin_relu(x_t, w["g2_r1"]), w["g2_r2"])
        ob = getattr(self, "_bufs", (None, None))[1]
        out32, _, relu_t = self.gated(x32, gate, r, E, want_relu=True, out=ob[:E] if ob is not None else None)
        return out32, relu_t

    def heads(self, relu_t):
        return self.lin(relu_t, self.weights()["heads"])          # [E,4]

    def heads_target_weight(self, relu_t, coords, wd, ht):
        """heads(relu_t) followed by target_weight, one launch (fp16 path); None when not available"""
        w = self.weights()
        if "heads_pack" not in w or not self.use_mlp or relu_t.dtype != torch.float16:
            return None
        E, P, dev = relu_t.shape[0], coords.shape[-1], relu_t.device
        target = torch.empty(1, E, 2, dtype=torch.float32, device=dev)
        weight = torch.empty(1, E, 2, dtype=torch.float32, device=dev)
        hwt, hb = w["heads_pack"]
        check(lib().ramp_upd_heads_linear(ptr(relu_t), ptr(hwt), ptr(hb), ptr(coords), ptr(target), ptr(weight), E, P,
                                          float(wd), float(ht), stream()), "ramp_upd_heads_linear")
        return target, weight

    def target_weight(self, hw, coords, wd, ht, want_delta=False):
        E = hw.shape[0]
        P = coords.shape[-1]
        dev = hw.device
        target = torch.empty(1, E, 2, dtype=torch.float32, device=dev)
        weight = torch.empty(1, E, 2, dtype=torch.float32, device=dev)
        delta = torch.empty(1, E, 2, dtype=torch.float32, device=dev) if want_delta else None
        check(lib().ramp_upd_heads(ptr(hw), ptr(coords), ptr(target), ptr(weight), ptr(delta), E, P, float(wd),
                                   float(ht), _code[self.dtype], stream()), "ramp_upd_heads")
        return target, weight, delta
