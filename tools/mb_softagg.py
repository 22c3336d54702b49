#!/usr/bin/env python
"""the fused SoftAgg launches alone (csrc/update_mlp.hip::upd_softagg_kernel / _finish) against the three launches they
replace, on a patch-like grouping (groups of ~19 factors scattered over the list) and a pair-like one (96 per group)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from rampvo_amd import ops
from rampvo_amd.synthetic import make_network
E = int(sys.argv[1]) if len(sys.argv) > 1 else 40000
net = make_network("SingleScale")
fu = net.update.fused(torch.float16)
w = fu.weights()
g = torch.Generator().manual_seed(1)
x32 = (torch.randn(E, 384, generator=g) * 0.5).cuda()
hy0 = (torch.randn(2200, 384, generator=g) * 0.5).half().cuda()
gid0 = torch.randint(0, 2200, (E,), generator=g).int().cuda()


def timed(fn, n=40):
    for _ in range(4):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3


for name, keys in (("patch-like", (torch.arange(E) // 19)[torch.randperm(E, generator=g)]), ("pair-like", torch.arange(E) // 96)):
    grp = ops.group_by(keys.cuda())
    G = int(grp.ngroups.item())
    from rampvo_amd._lib import check, lib, ptr, stream
    rows = lib().ramp_upd_softagg_frag_rows(E, G)
    frag = torch.empty(rows, 3, 384, device="cuda")
    hy = torch.empty(G, 384, dtype=torch.half, device="cuda")
    wf, bf, wg, bg = w["kk_fg_pack"]; wh, bh = w["kk_h_pack"]
    main = lambda: check(lib().ramp_upd_softagg(ptr(x32), ptr(hy0), ptr(gid0), ptr(grp.order), ptr(grp.gid), ptr(wf), ptr(bf), ptr(wg), ptr(bg), ptr(frag), E, stream()), "softagg")
    fin = lambda: check(lib().ramp_upd_softagg_finish(ptr(frag), ptr(grp.seg_start), ptr(grp.ngroups), ptr(wh), ptr(bh), ptr(hy), G, stream()), "finish")
    fg = torch.empty(E, 768, dtype=torch.half, device="cuda")
    old_fg = lambda: fu.fg(x32, hy0, gid0, w["kk_fg_pack"], E)
    fgv = old_fg()
    old_seg = lambda: fu.seg(fgv, grp, G)
    yv = old_seg()
    old_lin = lambda: fu.h_lin(yv, w["kk_h_pack"], grp)
    print("E=%d %-10s G=%5d | softagg %.1f + finish %.1f us | fg %.1f + softmax %.1f + linear %.1f us" %
          (E, name, G, timed(main), timed(fin), timed(old_fg), timed(old_seg), timed(old_lin)), flush=True)
