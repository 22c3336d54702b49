#!/usr/bin/env python
"""phase timeline of upd_gru_kernel (build with EXTRA=-DGRU_TRACE): wall_clock64 stamps of wave 0 of every workgroup"""
import os, sys, ctypes
sys.path.insert(0, "/root/repo")
import numpy as np, torch
exec(open("/root/repo/tools/mb_update.py").read().split("res = {}")[0])
from rampvo_amd._lib import lib
L = lib()._lib if hasattr(lib(), "_lib") else None
import rampvo_amd._lib as ll
so = ctypes.CDLL(os.path.join(os.path.dirname(ll.__file__), "csrc", "libramp_hip.so"))
for _ in range(3): gru()
torch.cuda.synchronize()
n = 32 * 640
buf = (ctypes.c_longlong * n)()
assert so.ramp_debug_gru_trace(buf, n) == 0
t = np.array(buf[:], dtype=np.int64).reshape(-1, 32)[:625]
names = {(0, 1): "stage-in (x32 -> res, pre-LN, Xs)"}
for st in (0, 1):
    o = 8 * st
    prev = 1 if st == 0 else 8
    names[(prev, 2 + o)] = "s%d gate gemm" % st
    names[(2 + o, 3 + o)] = "s%d sigmoid -> Gs" % st
    names[(3 + o, 4 + o)] = "s%d r1 gemm" % st
    names[(4 + o, 5 + o)] = "s%d relu -> Hs + barrier" % st
    names[(5 + o, 6 + o)] = "s%d r2 gemm" % st
    names[(6 + o, 7 + o)] = "s%d residual += g * r" % st
    names[(7 + o, 8 + o)] = "s%d %s" % (st, "LayerNorm -> Xs (2 barriers)" if st == 0 else "store out32 / relu_t")
print("wall_clock ticks are 10 ns; per-phase mean us over workgroups (first-round WGs: blocks < 256):")
for (i0, i1), nm in names.items():
    dd = (t[:, i1] - t[:, i0]) / 100.0
    print("  %-34s all %6.2f   first round %6.2f" % (nm, dd.mean(), dd[:256].mean()))
print("  total per WG                       all %6.2f   first round %6.2f" % ((t[:, 16] - t[:, 0]).mean() / 100, (t[:256, 16] - t[:256, 0]).mean() / 100))
print("  kernel span %.1f us; WG start times (percentiles, us): %s" % ((t[:, 16].max() - t[:, 0].min()) / 100, np.percentile((t[:, 0] - t[:, 0].min()) / 100, [0, 25, 50, 75, 100]).round(1)))
