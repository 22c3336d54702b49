#!/usr/bin/env python
"""phase breakdown of corr_mfma_kernel (library built with EXTRA=-DCORR_TRACE): shader cycles per edge (= per wave)
spent in setup (coords, union window), waiting for a batch's window loads (an explicit vmcnt(0) is inserted by the
trace build), MFMA + Cs stores, blend; run on a steady-state tracker's own graph."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from rampvo_amd import _lib
from rampvo_amd.config import make_cfg
from rampvo_amd.Ramp_vo import Ramp_vo
from rampvo_amd.synthetic import SyntheticStream, make_network
so = ctypes.CDLL(_lib.LIB_PATH)
MIXED = os.environ.get("CT_MIXED", "1") == "1"
slam = Ramp_vo(make_cfg("default", PATCHES_PER_FRAME=96, MIXED_PRECISION=MIXED), make_network("SingleScale"), {"event_bias": True})
slam.device_steps = False
st = SyntheticStream(480, 640, 60, seed=1234, device="cuda")
for t in range(60):
    im, ev, K, m = st.frame(t)
    slam(t, input_tensor=(ev, im, m), intrinsics=K)
plan = slam._graph_plan()
coords = slam.reproject()
torch.cuda.synchronize()
N = 5
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record()
for _ in range(N):
    slam.corr(coords, order=plan.g_ij.order)
e.record()
torch.cuda.synchronize()
E = len(slam._ii)
nw = min(-(-E // 8) * 8, 65536)
buf = (ctypes.c_ulonglong * (nw * 8))()
assert so.ramp_debug_corr_trace(buf, nw) == 0
rows = np.frombuffer(buf, dtype=np.uint64).reshape(nw, 8).astype(np.float64)
rows = rows[rows[:, 0] == 1]                                # workgroups that had an edge (the last launch)
v = rows.sum(0)
print("E = %d edges, %.1f us per launch (trace build: slower than the product kernel)" % (E, s.elapsed_time(e) * 1e3 / N))
names = ["waves", "total", "setup (coords, union)", "window loads (issue + wait)", "MFMA + Cs stores", "blend"]
for k in range(1, 6):
    print("  %-28s %8.0f cycles / edge  (%4.1f %%)" % (names[k], v[k] / v[0], 100 * v[k] / v[1]))
print("  other (A fragments, barriers, output stores) %8.0f cycles / edge" % ((v[1] - v[2:6].sum()) / v[0]))
live = rows[:, 7] == 1
print("  of which: indices + coordinates until the window geometry is known %8.0f cycles / edge" % (v[6] / v[0]))
for name, sel in (("live", live), ("dead (nothing in any plane)", ~live)):
    if sel.any():
        r = rows[sel]
        print("  %-28s %6d waves: total %8.0f cycles, preamble %6.0f, loads %6.0f, MFMA %6.0f, blend %6.0f, setup %6.0f" % (
            name, sel.sum(), r[:, 1].mean(), r[:, 6].mean(), r[:, 3].mean(), r[:, 4].mean(), r[:, 5].mean(), r[:, 2].mean()))
