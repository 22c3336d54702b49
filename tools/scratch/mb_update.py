"""micro-benchmark of the fused update-operator kernels at a fixed E (independent of tracker behaviour)"""
import os, sys, ctypes
sys.path.insert(0, '/root/repo')
import torch
from rampvo_amd.synthetic import make_network
from rampvo_amd.net import GraphPlan
from rampvo_amd._lib import lib, ptr, stream, check
E, M = int(os.environ.get("E", 40000)), 96
net = make_network("SingleScale")
fu = net.update.fused(torch.float16)
g = torch.Generator().manual_seed(3)
kk = torch.sort(torch.randint(0, M * 22, (E,), generator=g)).values.cuda()
ii = kk // M
jj = torch.randint(0, 22, (E,), generator=g).cuda()
plan = GraphPlan.build(ii, jj, kk)
x32 = (torch.randn(E, 384, generator=g) * 0.5).cuda()
w = fu.weights()
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3
out32 = torch.empty_like(x32); relu_t = torch.empty(E, 384, dtype=torch.float16, device="cuda"); tmp = torch.empty_like(x32)
_, _, wptr, bptr = w["gru_pack"]
def gru():
    check(lib().ramp_upd_gru(ptr(x32), None, None, None, None, 0.0, wptr, bptr, ptr(w["ln2"][0]), ptr(w["ln2"][1]), float(w["ln2"][2]), ptr(out32), ptr(relu_t), E, stream()), "gru")
wa, ba, wb, bb = w["c1_pack"]
def nbr():
    check(lib().ramp_upd_nbr(ptr(x32), ptr(plan.ix_raw), ptr(wa), ptr(ba), ptr(wb), ptr(bb), ptr(tmp), None, E, stream()), "nbr")
fgo = torch.empty(E, 768, dtype=torch.float16, device="cuda")
hyk = (torch.randn(2200, 384, generator=g) * 0.1).half().cuda()
gid = plan.g_kk.gid
wf, bf, wg, bg = w["kk_fg_pack"]
def fg_plain():
    check(lib().ramp_upd_fg(ptr(x32), None, None, None, ptr(wf), ptr(bf), ptr(wg), ptr(bg), ptr(fgo), E, stream()), "fg")
def fg_add():
    check(lib().ramp_upd_fg(ptr(x32), ptr(hyk), ptr(gid), ptr(tmp), ptr(wf), ptr(bf), ptr(wg), ptr(bg), ptr(fgo), E, stream()), "fg")
xh = x32.half()
def gemm768():
    return torch.nn.functional.linear(xh, w["kk_fg"][0], w["kk_fg"][1])
print("fg (no add) %.1f us   fg (+add, state write) %.1f us   hipBLASLt 384->768 GEMM %.1f us" % (timeit(fg_plain), timeit(fg_add), timeit(gemm768)))
xt = x32.half()
def gemm():
    return torch.nn.functional.linear(xt, w["c1a"][0], w["c1a"][1])
print("E=%d  gru %.1f us   nbr %.1f us   one hipBLASLt 384x384 GEMM %.1f us" % (E, timeit(gru), timeit(nbr), timeit(gemm)))
