#!/usr/bin/env python
"""phase timeline of upd_corr_tail_kernel<true> (build with EXTRA=-DGRU_TRACE)"""
import os, sys, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
exec(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "mb_update.py")).read().split("res = {}")[0])
import rampvo_amd._lib as ll
so = ctypes.CDLL(os.path.join(os.path.dirname(ll.__file__), "csrc", "libramp_hip.so"))
for _ in range(3): corr_mlp()
torch.cuda.synchronize()
n = 32 * 640
buf = (ctypes.c_longlong * n)()
assert so.ramp_debug_gru_trace(buf, n) == 0
t = np.array(buf[:], dtype=np.int64).reshape(-1, 32)[:625]
names = ["Linear1 over K=896 (3 staged chunks)", "Linear2 gemm", "bias -> y, barrier, LN row pass x2, h -> tile", "Linear3 gemm", "row pass 2: net + inp + c, LN, store"]
for k, nm in enumerate(names):
    d = (t[:, k + 1] - t[:, k]) / 100.0
    print("  %-52s %6.2f us" % (nm, d.mean()))
print("  total per WG %.2f us; kernel span %.1f us; WG starts (percentiles): %s" % (((t[:, 5] - t[:, 0]) / 100).mean(), (t[:, 5].max() - t[:, 0].min()) / 100, np.percentile((t[:, 0] - t[:, 0].min()) / 100, [0, 50, 75, 90, 100]).round(1)))
