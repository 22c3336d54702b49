#!/usr/bin/env python
"""microbenchmark of the fp32-accuracy update-operator chains (csrc/update_x3.hip) at the bench size (HIP events, 30 launches
each), next to the library-GEMM operator they replace (RAMP_X3=0's path) -- tools/mb_update.py is the fp16 twin"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
from rampvo_amd._lib import check, lib, ptr, stream
from rampvo_amd.synthetic import make_network
E = int(sys.argv[1]) if len(sys.argv) > 1 else 40000
net = make_network("SingleScale")
fu = net.update.fused(torch.float32)
w = fu.weights()
g = torch.Generator().manual_seed(1)
rnd = lambda *s: (torch.randn(*s, generator=g) * 0.5).cuda()
G = 2200
x32, h0, h1 = rnd(E, 384), rnd(G, 384), rnd(G, 384)
gid = (torch.arange(E) // 19).int().cuda() % G
order = torch.argsort(gid.long(), stable=True).int()
seg = torch.zeros(G + 1, dtype=torch.int32, device="cuda")
seg[1:] = torch.cumsum(torch.bincount(gid.long(), minlength=G), 0).int()
ng = torch.tensor([G], dtype=torch.int32, device="cuda")
out32 = torch.empty(E, 384, device="cuda")
idx = (torch.arange(E) - 1).cuda()
corr = F.pad(rnd(E, 882), (0, 14)).contiguous()
net_map = torch.randint(-1, E, (E,), generator=g).cuda()
table = rnd(3072, 384); inp_idx = torch.randint(0, 100000, (E,), generator=g).cuda()
fg = torch.empty(E, 768, device="cuda")
y, hy = torch.empty(G, 384, device="cuda"), torch.zeros(G, 384, device="cuda")
coords = (torch.rand(E, 2, 3, 3, generator=g) * 60).cuda()
target, weight = torch.empty(1, E, 2, device="cuda"), torch.empty(1, E, 2, device="cuda")
_, _, wptr, bptr = w["gru_pack"]; ln1, ln2 = w["ln1"], w["ln2"]
hw, hb = w["heads_pack"]
def gru():
    check(lib().ramp_x3_gru(ptr(x32), ptr(h0), ptr(gid), ptr(h1), ptr(gid), ptr(ln1[0]), ptr(ln1[1]), float(ln1[2]), wptr, bptr,
                            ptr(ln2[0]), ptr(ln2[1]), float(ln2[2]), ptr(out32), None, E, ptr(hw), ptr(hb), ptr(coords),
                            ptr(target), ptr(weight), 3, 160.0, 120.0, stream()), "gru")
wa, ba, wb, bb = w["c1_pack"]
def nbr():
    check(lib().ramp_x3_nbr(ptr(x32), ptr(idx), ptr(wa), ptr(ba), ptr(wb), ptr(bb), ptr(out32), E, stream()), "nbr")
w1, b1 = w["corr1_pack"]; w2, b2, w3, b3 = w["tail_pack"]; ln, nm = w["corr_ln"], w["norm"]
def corr_mlp():
    check(lib().ramp_x3_corr_mlp(ptr(corr), 896, ptr(w1), ptr(b1), ptr(w2), ptr(b2), ptr(w3), ptr(b3), ptr(ln[0]), ptr(ln[1]),
                                 float(ln[2]), ptr(x32), ptr(net_map), ptr(table), ptr(inp_idx), 3072, ptr(nm[0]), ptr(nm[1]),
                                 float(nm[2]), ptr(out32), E, stream()), "corr_mlp")
wf, bf, wg, bg = w["kk_fg_pack"]
def fgk():
    check(lib().ramp_x3_fg(ptr(x32), ptr(h0), ptr(gid), None, ptr(wf), ptr(bf), ptr(wg), ptr(bg), ptr(fg), E, stream()), "fg")
def segk():
    check(lib().ramp_x3_segment_softmax(ptr(fg), ptr(order), ptr(seg), ptr(ng), ptr(y), G, stream()), "seg")
hp = w["kk_h_pack"]
def lin():
    check(lib().ramp_x3_linear(ptr(y), ptr(hp[0]), ptr(hp[1]), ptr(hy), G, ptr(ng), stream()), "lin")
def lib_gemm():                      # one [E,384] x [384,384] fp32 library GEMM: what a layer of RAMP_X3=0's path costs
    torch.mm(x32, w["c1a"][0].t(), out=out32)
res = {}
for name, fn in (("gru+heads", gru), ("nbr", nbr), ("corr_mlp", corr_mlp), ("fg", fgk), ("segsoftmax", segk), ("h", lin),
                 ("torch.mm fp32 (1 layer)", lib_gemm)):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(30):
        fn()
    e.record(); torch.cuda.synchronize()
    res[name] = s.elapsed_time(e) / 30 * 1e3
op = res["corr_mlp"] + 2 * res["nbr"] + 2 * (res["fg"] + res["segsoftmax"] + res["h"]) + res["gru+heads"]
print("E=%d x3: " % E + "  ".join("%s %.1f us" % kv for kv in res.items()) + "  | operator %.0f us = %.0f TFLOP/s fp32-equivalent"
      % (op, E * 5.4e6 / op / 1e6), flush=True)
