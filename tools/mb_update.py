#!/usr/bin/env python
"""microbenchmark of the fused update-operator chains at the bench size (HIP events, 50 launches each)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
from rampvo_amd._lib import check, lib, ptr, stream
from rampvo_amd.synthetic import make_network
E = int(sys.argv[1]) if len(sys.argv) > 1 else 40000
net = make_network("SingleScale")
fu = net.update.fused(torch.float16)
w = fu.weights()
g = torch.Generator().manual_seed(1)
rnd = lambda *s: (torch.randn(*s, generator=g) * 0.5).cuda()
x32, hy = rnd(E, 384), rnd(2200, 384).half()
gid = torch.randint(0, 2200, (E,), generator=g).int().cuda()
out32 = torch.empty(E, 384, device="cuda"); relu_t = torch.empty(E, 384, dtype=torch.half, device="cuda")
idx = (torch.arange(E) - 1).cuda() if os.environ.get("MB_ADJ") else torch.randint(-1, E, (E,), generator=g).cuda()
if os.environ.get("MB_ADJ"):
    gid = (torch.arange(E) // 19).int().cuda()
corr = F.pad(rnd(E, 882).half(), (0, 14)).contiguous()
net_map = torch.randint(-1, E, (E,), generator=g).cuda()
table = rnd(3072, 384).half(); inp_idx = torch.randint(0, 100000, (E,), generator=g).cuda()
fg = torch.empty(E, 768, dtype=torch.half, device="cuda")
_, _, wptr, bptr = w["gru_pack"]; ln1, ln2 = w["ln1"], w["ln2"]
def gru():
    check(lib().ramp_upd_gru(ptr(x32), ptr(hy), ptr(gid), ptr(ln1[0]), ptr(ln1[1]), float(ln1[2]), wptr, bptr,
                             ptr(ln2[0]), ptr(ln2[1]), float(ln2[2]), ptr(out32), ptr(relu_t), E, stream()), "gru")
wa, ba, wb, bb = w["c1_pack"]
def nbr():
    check(lib().ramp_upd_nbr(ptr(x32), ptr(idx), ptr(wa), ptr(ba), ptr(wb), ptr(bb), ptr(out32), None, E, stream()), "nbr")
w1, b1 = w["corr1_pack"]; w2, b2, w3, b3 = w["tail_pack"]; ln, nm = w["corr_ln"], w["norm"]
def corr_mlp():
    check(lib().ramp_upd_corr_mlp(ptr(corr), 896, ptr(w1), ptr(b1), ptr(w2), ptr(b2), ptr(w3), ptr(b3), ptr(ln[0]),
                                  ptr(ln[1]), float(ln[2]), ptr(x32), ptr(net_map), ptr(table), ptr(inp_idx), 3072,
                                  ptr(nm[0]), ptr(nm[1]), float(nm[2]), ptr(out32), E, stream()), "corr_mlp")
wf, bf, wg, bg = w["kk_fg_pack"]
def fgk():
    check(lib().ramp_upd_fg(ptr(x32), ptr(hy), ptr(gid), None, ptr(wf), ptr(bf), ptr(wg), ptr(bg), ptr(fg), E, stream()), "fg")
res = {}
for name, fn in (("gru", gru), ("nbr", nbr), ("corr_mlp", corr_mlp), ("fg", fgk)):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(50):
        fn()
    e.record(); torch.cuda.synchronize()
    res[name] = s.elapsed_time(e) / 50 * 1e3
print("E=%d RAMP_UPD_BIG=%s: " % (E, os.environ.get("RAMP_UPD_BIG", "auto")) + "  ".join("%s %.1f us" % kv for kv in res.items()), flush=True)
