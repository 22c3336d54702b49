#!/usr/bin/env python
"""Per-section wall time of a steady-state tracker step (synchronised around every section, so
sections include their host overhead; the sum exceeds the pipelined step time)."""
import os, sys, time, collections
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rampvo_amd.config import make_cfg
from rampvo_amd.Ramp_vo import Ramp_vo
from rampvo_amd.synthetic import SyntheticStream, make_network
from rampvo_amd import Ramp_vo as RV, ops, altcorr, fastba, projective_ops as pops, net as NET

mixed = int(os.environ.get("MIXED", "1"))
SYNC = int(os.environ.get("SYNC", "1"))     # SYNC=0: host-side (enqueue) time per section only
acc = collections.defaultdict(float); cnt = collections.defaultdict(int)
enabled = [False]
def timed(name, fn):
    def w(*a, **k):
        if not enabled[0]:
            return fn(*a, **k)
        if SYNC: torch.cuda.synchronize()
        t = time.perf_counter()
        r = fn(*a, **k)
        if SYNC: torch.cuda.synchronize()
        acc[name] += time.perf_counter() - t; cnt[name] += 1
        return r
    return w

cfg = make_cfg("default", PATCHES_PER_FRAME=96, MIXED_PRECISION=bool(mixed))
netw = make_network("SingleScale")
slam = Ramp_vo(cfg, netw, {"event_bias": True})
netw.patchify.forward = timed("patchify(total)", netw.patchify.forward)
netw.patchify.encoder.forward = timed("  encoder", netw.patchify.encoder.forward)
slam.update = timed("update(total)", slam.update)
slam.keyframe = timed("keyframe(total)", slam.keyframe)
slam.append_factors = timed("append_factors", slam.append_factors)
slam._graph_plan = timed("  graph_plan", slam._graph_plan)
slam.reproject = timed("  reproject", slam.reproject)
slam.corr = timed("  corr", slam.corr)
RV.fastba.BA = timed("  BA", RV.fastba.BA)
RV.pops.point_cloud = timed("  point_cloud", RV.pops.point_cloud)
RV.pops.flow_mag = timed("  flow_mag", RV.pops.flow_mag)
slam._motionmag_pair = timed("  motionmag+sync", slam._motionmag_pair)
slam._track = timed("track(total)", slam._track)
import rampvo_amd.update_fused as UF
UF.FusedUpdate.hidden = timed("  fused_hidden", UF.FusedUpdate.hidden)
import rampvo_amd.utils as U
NET.get_coords_from_topk_events = timed("  topk_coords", NET.get_coords_from_topk_events)

T = 110
stream = SyntheticStream(480, 640, T + 1, seed=1234, device="cuda")
frames = [stream.frame(t) for t in range(T)]
for t in range(T):
    if t == 80:
        enabled[0] = True; torch.cuda.synchronize(); t0 = time.perf_counter()
    im, ev, K, mask = frames[t]
    slam(t, input_tensor=(ev, im, mask), intrinsics=K)
torch.cuda.synchronize(); total = time.perf_counter() - t0
n = T - 80
print("steps %d, E=%d, %s step time %.3f ms" % (n, len(slam._ii), "synchronised" if SYNC else "pipelined (host enqueue times below)", 1e3 * total / n))
for k in sorted(acc, key=lambda k: -acc[k]):
    print("%-22s %7.3f ms/step  (%d calls)" % (k, 1e3 * acc[k] / n, cnt[k]))
