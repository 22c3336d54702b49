#!/usr/bin/env python
"""who is the main stream waiting for at the top of a device-resident step?  Per steady-state frame: time from the gate
(recorded inside ramp_track_step before the gru chain) to (a) the end of the step's own tail and (b) the end of the next
frame's front end, which was launched behind that gate -- the next step starts at max(a, b)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from rampvo_amd.config import make_cfg
from rampvo_amd.Ramp_vo import Ramp_vo
from rampvo_amd.synthetic import SyntheticStream, make_network
from rampvo_amd import track_dev
N0, N = 120, 100
slam = Ramp_vo(make_cfg("default", PATCHES_PER_FRAME=96, MIXED_PRECISION=True), make_network("SingleScale"), {"event_bias": True})
slam.inputs_ready = True
st = SyntheticStream(480, 640, N0 + N + 1, seed=1234, device="cuda")
frames = [st.frame(t) for t in range(N0 + N)]
torch.cuda.synchronize()
mk = lambda: torch.cuda.Event(enable_timing=True)
rec = []
inner = track_dev.DeviceTrack.step
os.environ["RAMP_GATE_FLAG"] = "0"       # the gate must be an event here: it is the time origin of both chains
def step(dv, counter, flags, k_new=None, gate_event=None, **kw):
    g, e = mk(), mk()
    g.record(); e.record()            # create the handles
    dv.t.probe[2] = None
    inner(dv, counter, flags, k_new=k_new, gate_event=g.cuda_event)     # our own (timing) event as the gate
    slam._ev_gate = g
    e.record()
    rec.append([g, e, None])
track_dev.DeviceTrack.step = step
pat = slam.network.patchify
run_inner = pat._run_graph
def run(graph):
    run_inner(graph)
    f = mk(); f.record()
    if rec:
        rec[-1][2] = f
pat._run_graph = run
for t in range(N0 + N):
    im, ev, K, m = frames[t]
    slam(t, input_tensor=(ev, im, m), intrinsics=K)
torch.cuda.synchronize()
tail = np.array([g.elapsed_time(e) for g, e, f in rec[-N:-1] if f is not None]) * 1e3
fe = np.array([g.elapsed_time(f) for g, e, f in rec[-N:-1] if f is not None]) * 1e3
print("gate -> end of the step's tail (gru, heads, BA, motion test, edit, plan): %.0f us (p10 %.0f, p90 %.0f)" % (tail.mean(), np.percentile(tail, 10), np.percentile(tail, 90)))
print("gate -> end of the next front end (selection + graph):                    %.0f us (p10 %.0f, p90 %.0f)" % (fe.mean(), np.percentile(fe, 10), np.percentile(fe, 90)))
print("front end later than the tail by %.0f us on average; later in %.0f %% of the frames" % ((fe - tail).mean(), 100 * (fe > tail).mean()))
