#!/usr/bin/env python
"""can the correlation launch and the corr-MLP launch behind it run side by side?  The tracker's own steady-state factors,
re-ordered so that schedule position == row: (a) correlation over all rows, then corr-MLP over all rows (the frame's
order today); (b) NC chunks of rows -- correlation chunk c on one stream, corr-MLP chunk c on a second stream behind an
event, so chunk c's MLP runs next to chunk c+1's correlation.  Warm and behind 1.5 GB of other traffic."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from rampvo_amd import _lib
from rampvo_amd._lib import check, lib, ptr
from rampvo_amd.config import make_cfg
from rampvo_amd.Ramp_vo import Ramp_vo, RAMP_NHWC32
CORR_ROW = 896
from rampvo_amd.synthetic import SyntheticStream, make_network
torch.manual_seed(1234)
net = make_network("SingleScale")
slam = Ramp_vo(make_cfg("default", PATCHES_PER_FRAME=96, MIXED_PRECISION=True), net, {"event_bias": True})
T = int(os.environ.get("T", 120))
st = SyntheticStream(480, 640, T + 1, seed=1234, device="cuda")
with torch.no_grad():
    for t in range(T):
        im, ev, K, mask = st.frame(t)
        slam(t, input_tensor=(ev, im, mask), intrinsics=K)
torch.cuda.synchronize()
dv = slam._dev
E = int(dv.dyn.cpu()[2])
g = dv.graph[dv.cur]
o = dv.ij["order"][:E].long()
jj, kk = g[1, :E][o].contiguous(), g[2, :E][o].contiguous()
co = dv.coords[:E][o].contiguous()
print("E =", E, flush=True)
slam._corr_launch(co[None], kk, jj, None)          # (builds the level table)
lv = slam._corr_levels
corr = torch.zeros(E, CORR_ROW, dtype=torch.half, device="cuda")
fu = net.update.fused(torch.float16)
w = fu.weights()
gen = torch.Generator().manual_seed(1)
rnd = lambda *s: (torch.randn(*s, generator=gen) * 0.5).cuda()
x32 = rnd(E, 384); out32 = torch.empty(E, 384, device="cuda")
net_map = torch.randint(-1, E, (E,), generator=gen).cuda()
table = rnd(3072, 384).half(); inp_idx = torch.randint(0, 100000, (E,), generator=gen).cuda()
w1, b1 = w["corr1_pack"]; w2, b2, w3, b3 = w["tail_pack"]; ln, nm = w["corr_ln"], w["norm"]


def corr_rows(r0, n):
    check(lib().ramp_corr_fwd_ordered(
        ptr(slam.gmap_), lv, 2, co.data_ptr() + r0 * 72, kk.data_ptr() + r0 * 8, jj.data_ptr() + r0 * 8, None,
        corr.data_ptr() + r0 * CORR_ROW * 2, CORR_ROW, slam.M * slam.mem, slam.mem, n, slam.mem * slam.M, slam.mem, 128, 3, 3,
        _lib.RAMP_F16, RAMP_NHWC32, _lib.stream()), "corr")


def mlp_rows(r0, n):
    check(lib().ramp_upd_corr_mlp(corr.data_ptr() + r0 * CORR_ROW * 2, 896, ptr(w1), ptr(b1), ptr(w2), ptr(b2), ptr(w3), ptr(b3),
                                  ptr(ln[0]), ptr(ln[1]), float(ln[2]), ptr(x32), net_map.data_ptr() + r0 * 8, ptr(table),
                                  inp_idx.data_ptr() + r0 * 8, 3072, ptr(nm[0]), ptr(nm[1]), float(nm[2]),
                                  out32.data_ptr() + r0 * 1536, n, _lib.stream()), "corr_mlp")


s2 = torch.cuda.Stream()
big = torch.empty(768 * 1024 * 1024 // 4, device="cuda")


def serial():
    corr_rows(0, E); mlp_rows(0, E)


def chunked(nc, align=64):
    step = (E + nc - 1) // nc
    step = (step + align - 1) // align * align
    s1 = torch.cuda.current_stream()
    r0 = 0
    while r0 < E:
        n = min(step, E - r0)
        corr_rows(r0, n)
        ev = torch.cuda.Event(); ev.record(s1)
        s2.wait_event(ev)
        with torch.cuda.stream(s2):
            mlp_rows(r0, n)
        r0 += n
    s1.wait_stream(s2)


def chunked_one_stream(nc, align=64):
    step = (E + nc - 1) // nc
    step = (step + align - 1) // align * align
    for r0 in range(0, E, step):
        corr_rows(r0, min(step, E - r0))
    for r0 in range(0, E, step):
        mlp_rows(r0, min(step, E - r0))


from rampvo_amd.track_dev import Signal
sig = Signal()
seq = [0]


def chunked_flag(nc, align=64):
    """the cross-stream go as a word in signal memory (one storing thread / one sleeping wave), not an event"""
    step = (E + nc - 1) // nc
    step = (step + align - 1) // align * align
    s1 = torch.cuda.current_stream()
    for r0 in range(0, E, step):
        n = min(step, E - r0)
        corr_rows(r0, n)
        seq[0] += 1
        check(lib().ramp_stream_signal(_lib.stream(), sig.ptr, seq[0]), "signal")
        sig.wait(s2, seq[0])
        with torch.cuda.stream(s2):
            mlp_rows(r0, n)
    seq[0] += 1
    with torch.cuda.stream(s2):
        check(lib().ramp_stream_signal(_lib.stream(), sig.ptr, seq[0]), "signal")
    sig.wait(s1, seq[0])


def timed(fn, cold, n=15):
    for _ in range(3):
        fn()
    evs = []
    for _ in range(n):
        if cold:
            big.add_(1.0)
        check(lib().ramp_stream_delay(1500, _lib.stream()), "delay")     # the host enqueues everything behind a sleeping wave
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record()
        evs.append((a, b))
    torch.cuda.synchronize()
    return 1e3 * float(np.median([a.elapsed_time(b) for a, b in evs]))


serial(); ref = out32.clone(); torch.cuda.synchronize()
for cold in (False, True):
    print("== %s" % ("behind 1.5 GB of other traffic" if cold else "warm"))
    print("  correlation alone   %6.1f us" % timed(lambda: corr_rows(0, E), cold))
    print("  corr-MLP alone      %6.1f us" % timed(lambda: mlp_rows(0, E), cold))
    print("  one behind the other %5.1f us" % timed(serial, cold))
    for nc in (2, 3, 4, 6, 8):
        res = []
        for f in (chunked_one_stream, chunked, chunked_flag):
            out32.zero_()
            t = timed(lambda: f(nc), cold)
            torch.cuda.synchronize()
            res.append("%5.1f us%s" % (t, "" if torch.equal(out32, ref) else " (DIFFERENT ROWS)"))
        print("  %2d chunks: one stream %s, two streams + events %s, two streams + signal words %s" % (nc, *res), flush=True)
print("== how the two launches scale with the row count (warm; rows from the middle of the schedule)")
for n in (E, E // 2, E // 4, E // 8, 2048, 512, 64):
    r0 = ((E - n) // 2) // 64 * 64
    print("  %6d rows: correlation %6.1f us, corr-MLP %6.1f us" % (n, timed(lambda: corr_rows(r0, n), False), timed(lambda: mlp_rows(r0, n), False)), flush=True)
# the slowest single factors
cx = co[:, 0].reshape(E, 9); cy = co[:, 1].reshape(E, 9)
ok = torch.isfinite(cx).all(1) & torch.isfinite(cy).all(1) & (cx.abs().amax(1) < 1e6) & (cy.abs().amax(1) < 1e6)
area = torch.where(ok, (cx.amax(1) - cx.amin(1) + 8) * (cy.amax(1) - cy.amin(1) + 8), torch.zeros(E, device="cuda"))
inside = ((cx > -4) & (cx < 164) & (cy > -4) & (cy < 124)).any(1) & ok
print("  factors with a union window above 192 positions at level 0: %d of %d (%d of them touch the plane)" % (
    int((area > 192).sum()), E, int(((area > 192) & inside).sum())))
big_ix = torch.nonzero((area > 192) & inside).flatten()
if big_ix.numel():
    for k in big_ix[:3].tolist() + big_ix[-3:].tolist():
        print("    factor %6d (area %8.0f): correlation launch of that one row %5.1f us" % (k, float(area[k]), timed(lambda: corr_rows(k, 1), False)))
small_ix = torch.nonzero((area <= 192) & inside).flatten()
for k in small_ix[:3].tolist():
    print("    factor %6d (area %8.0f): correlation launch of that one row %5.1f us" % (k, float(area[k]), timed(lambda: corr_rows(k, 1), False)))
