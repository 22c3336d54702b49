"""GPU-side time from the end of frame t's motion test to the start of frame t+1's correlation kernel (main stream),
and from there to the next motion test; usage: gap3.py [pipeline 0|1]"""
import sys, time; sys.path.insert(0, '/root/repo')
import torch
from rampvo_amd import ops
from rampvo_amd.config import make_cfg
from rampvo_amd.Ramp_vo import Ramp_vo
from rampvo_amd.synthetic import SyntheticStream, make_network
cfg = make_cfg("default", PATCHES_PER_FRAME=96, MIXED_PRECISION=True)
slam = Ramp_vo(cfg, make_network("SingleScale"), {"event_bias": True})
slam.inputs_ready = len(sys.argv) > 1 and sys.argv[1] == "1"
T = 200
stream = SyntheticStream(480, 640, T + 1, seed=1234, device="cuda")
frames = [stream.frame(t) for t in range(T)]
ev_mm, ev_corr = [], []
on = [False]
mm0 = ops.motionmag
def mm(*a, **k):
    r = mm0(*a, **k)
    if on[0]:
        e = torch.cuda.Event(enable_timing=True); e.record(); ev_mm.append(e)
    return r
ops.motionmag = mm
c0 = slam.corr
def corr(*a, **k):
    if on[0] and k.get("order") is not None:
        e = torch.cuda.Event(enable_timing=True); e.record(); ev_corr.append(e)
    return c0(*a, **k)
slam.corr = corr
flags = []
fin0 = slam._keyframe_finish
def fin():
    pend = slam._pending
    n0 = slam.n
    r = fin0()
    if on[0]: flags.append((slam.n < n0, len(pend["spec"]) and (slam.n < n0) not in pend["spec"]))
    return r
slam._keyframe_finish = fin
for t in range(T):
    if t == 80:
        torch.cuda.synchronize(); on[0] = True; t0 = time.perf_counter()
    im, ev, K, mask = frames[t]; slam(t, input_tensor=(ev, im, mask), intrinsics=K)
slam.settle(); torch.cuda.synchronize(); dt = time.perf_counter() - t0
n = T - 80
m = min(len(ev_mm) - 1, len(ev_corr) - 1)
gaps = [ev_mm[i].elapsed_time(ev_corr[i + 1]) for i in range(m)]
work = [ev_corr[i].elapsed_time(ev_mm[i]) for i in range(min(len(ev_mm), len(ev_corr)))]
import statistics as st
print("pipeline %d: step %.1f us; motion test -> next corr start %.1f us (median %.1f); corr start -> motion test %.1f us"
      % (slam.inputs_ready, 1e6 * dt / n, 1e3 * st.mean(gaps), 1e3 * st.median(gaps), 1e3 * st.mean(work)))

import numpy as np
g = np.array(gaps) * 1e3; f = np.array(flags[:len(g)])
for name, sel in (("removed", f[:, 0]), ("kept", ~f[:, 0]), ("mispredicted", f[:, 1]), ("predicted", ~f[:, 1])):
    if sel.any(): print("  %-13s n=%3d  gap mean %.0f median %.0f p90 %.0f" % (name, sel.sum(), g[sel].mean(), np.median(g[sel]), np.percentile(g[sel], 90)))
print("  sorted gaps:", np.sort(g).astype(int)[::6])
