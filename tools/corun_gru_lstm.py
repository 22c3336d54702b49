#!/usr/bin/env python
"""does the front end's LSTM launch run BESIDE the update operator's gru launch or in turns with it?  gru alone, the LSTM
step alone, and both started together on two streams (HIP events on the gru stream / around both)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from rampvo_amd._lib import check, lib, ptr, stream
from rampvo_amd.synthetic import make_network
from rampvo_amd import conv_hip
E = 40000
net = make_network("SingleScale")
fu = net.update.fused(torch.float16)
w = fu.weights()
g = torch.Generator().manual_seed(1)
rnd = lambda *s: (torch.randn(*s, generator=g) * 0.5).cuda()
x32, hy = rnd(E, 384), rnd(2200, 384).half()
gid = torch.randint(0, 2200, (E,), generator=g).int().cuda()
out32 = torch.empty(E, 384, device="cuda"); relu_t = torch.empty(E, 384, dtype=torch.half, device="cuda")
_, _, wptr, bptr = w["gru_pack"]; ln1, ln2 = w["ln1"], w["ln2"]
def gru():
    check(lib().ramp_upd_gru(ptr(x32), ptr(hy), ptr(gid), ptr(ln1[0]), ptr(ln1[1]), float(ln1[2]), wptr, bptr,
                             ptr(ln2[0]), ptr(ln2[1]), float(ln2[2]), ptr(out32), ptr(relu_t), E, stream()), "gru")
enc = net.patchify.encoder
H, W = 480, 640
ev, im = rnd(5, H, W), rnd(3, H, W)
st = conv_hip.LstmState(H * W, "cuda")
def lstm():
    conv_hip.lstm_superstate_step(enc, ev, im, st)
s2 = torch.cuda.Stream()
def timed(fa, fb=None, n=30):
    for _ in range(3):
        fa()
        if fb:
            with torch.cuda.stream(s2): fb()
    torch.cuda.synchronize()
    tot = 0.0
    for _ in range(n):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        a.record()
        if fb:
            s2.wait_event(a)
            with torch.cuda.stream(s2): fb()
        fa()
        if fb:
            torch.cuda.current_stream().wait_stream(s2)
        b.record(); torch.cuda.synchronize()
        tot += a.elapsed_time(b)
    return tot / n * 1e3
def both(order):
    res = []
    for _ in range(3):
        gru(); lstm()
    torch.cuda.synchronize()
    for _ in range(20):
        a, eg, el = (torch.cuda.Event(enable_timing=True) for _ in range(3))
        torch.cuda.synchronize()
        a.record()
        s2.wait_event(a)
        if order == "lstm first":
            with torch.cuda.stream(s2): lstm(); el.record()
            gru(); eg.record()
        else:
            gru(); eg.record()
            with torch.cuda.stream(s2): lstm(); el.record()
        torch.cuda.synchronize()
        res.append((a.elapsed_time(eg) * 1e3, a.elapsed_time(el) * 1e3))
    return [sum(r[i] for r in res) / len(res) for i in range(2)]
big = torch.empty(48 * 1024 * 1024, device="cuda")
_lstm_full = lstm
variants = {"LSTM step": _lstm_full, "torch add_ on 192 MB": lambda: big.add_(1.0), "a second gru": gru}
for name, fn in variants.items():
    lstm = fn
    for order in ("gru first", "lstm first"):
        g_end, l_end = both(order)
        print("co-runner %-22s (%s): gru done after %.1f us, co-runner after %.1f us" % (name, order.replace("lstm", "co-runner"), g_end, l_end))
lstm = _lstm_full
# the LSTM step arriving d small launches (~5 us each) after gru has started, as in the frame (the selection runs first)
tiny = torch.zeros(64, device="cuda")
for d in (0, 2, 4, 8, 12, 16, 24):
    def delayed(d=d):
        for _ in range(d):
            tiny.add_(1.0)
        _lstm_full()
    lstm = delayed
    g_end, l_end = both("gru first")
    print("LSTM step %2d small launches behind the gru launch: gru done after %.1f us, LSTM after %.1f us" % (d, g_end, l_end))
lstm = _lstm_full
# the same with the gru launch on a HIGH-priority stream (the LSTM's stream keeps the default priority)
hi = torch.cuda.Stream(priority=-1)
for d in (0, 4, 8):
    def delayed(d=d):
        for _ in range(d):
            tiny.add_(1.0)
        _lstm_full()
    lstm = delayed
    with torch.cuda.stream(hi):
        g_end, l_end = both("gru first")
    print("high-priority gru, LSTM step %2d small launches behind: gru done after %.1f us, LSTM after %.1f us" % (d, g_end, l_end))
lstm = _lstm_full
for order in ():
    g_end, l_end = both(order)
    print("launched together (%s): gru done after %.1f us, LSTM step done after %.1f us" % (order, g_end, l_end))
print("gru alone %.1f us | LSTM step alone %.1f us | both started together, until both are done %.1f us" % (timed(gru), timed(lstm), timed(gru, lstm)))
