"""host time between the keyframe read-back returning and the next correlation launch, by stage (pipelined mode)"""
import sys, time, collections; sys.path.insert(0, '/root/repo')
import torch
from rampvo_amd import ops, fastba
from rampvo_amd import projective_ops as pops
from rampvo_amd.config import make_cfg
from rampvo_amd.Ramp_vo import Ramp_vo
from rampvo_amd.synthetic import SyntheticStream, make_network
cfg = make_cfg("default", PATCHES_PER_FRAME=96, MIXED_PRECISION=True)
net = make_network("SingleScale")
slam = Ramp_vo(cfg, net, {"event_bias": True})
slam.inputs_ready = True
T = 200
stream = SyntheticStream(480, 640, T + 1, seed=1234, device="cuda")
frames = [stream.frame(t) for t in range(T)]
on = [False]; inwin = [False]; t_sync = [0.0]
acc = collections.Counter(); cnt = collections.Counter()
def wrap(obj, name, label=None):
    f = getattr(obj, name); label = label or name
    def g(*a, **k):
        t = time.perf_counter(); r = f(*a, **k)
        if on[0] and inwin[0]: acc[label] += time.perf_counter() - t; cnt[label] += 1
        return r
    setattr(obj, name, g)
s0 = torch.cuda.Event.synchronize
def sync(self):
    r = s0(self)
    if on[0]: inwin[0] = True; t_sync[0] = time.perf_counter()
    return r
torch.cuda.Event.synchronize = sync
c0 = slam.corr
def corr(*a, **k):
    r = c0(*a, **k)
    if on[0] and inwin[0] and k.get("order") is not None:
        acc["TOTAL sync->corr launched"] += time.perf_counter() - t_sync[0]; cnt["TOTAL sync->corr launched"] += 1; inwin[0] = False
    return r
slam.corr = corr
for n in ("_spec_outcome", "_build_next_plan", "_apply_removal", "_prefetch_edges", "append_factors", "reproject",
          "_initial_depth", "_graph_plan", "_upload"):
    wrap(slam, n)
for n in ("frame_begin", "depth_median_fill", "store_rows"):
    wrap(ops, n)
wrap(slam, "_corr_launch"); wrap(slam, "_keyframe_finish"); wrap(slam, "_wait_upload_stream")
import rampvo_amd.ops as _o
_sp = _o.ShiftPlan.run
def _run(self, k, n):
    t = time.perf_counter(); r = _sp(self, k, n)
    if on[0] and inwin[0]: acc["shift_rows"] += time.perf_counter() - t; cnt["shift_rows"] += 1
    return r
_o.ShiftPlan.run = _run
wrap(torch.cuda.Stream, "wait_stream"); wrap(torch.cuda.Stream, "wait_event"); wrap(torch.cuda.Event, "record")
wrap(torch.cuda, "current_stream")
for t in range(T):
    if t == 80:
        torch.cuda.synchronize(); on[0] = True
    im, ev, K, mask = frames[t]; slam(t, input_tensor=(ev, im, mask), intrinsics=K)
slam.settle(); torch.cuda.synchronize()
n = cnt["TOTAL sync->corr launched"]
for k, v in acc.most_common():
    print("  %-22s %7.1f us/frame  %5.2f calls/frame" % (k, 1e6 * v / n, cnt[k] / n))
