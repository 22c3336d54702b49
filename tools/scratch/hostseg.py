"""wall-clock host time per frame spent inside the tracker's stages (perf_counter wrappers, no profiler);
usage: hostseg.py [pipeline 0|1]"""
import sys, time, collections; sys.path.insert(0, '/root/repo')
import torch
from rampvo_amd import fastba, ops, altcorr
from rampvo_amd import projective_ops as pops
from rampvo_amd.config import make_cfg
from rampvo_amd.Ramp_vo import Ramp_vo
from rampvo_amd.synthetic import SyntheticStream, make_network
cfg = make_cfg("default", PATCHES_PER_FRAME=96, MIXED_PRECISION=True)
net = make_network("SingleScale")
slam = Ramp_vo(cfg, net, {"event_bias": True})
slam.inputs_ready = len(sys.argv) > 1 and sys.argv[1] == "1"
acc = collections.Counter(); cnt = collections.Counter(); on = [False]
def wrap(obj, name, label=None):
    f = getattr(obj, name); label = label or name
    def g(*a, **k):
        t = time.perf_counter(); r = f(*a, **k)
        if on[0]: acc[label] += time.perf_counter() - t; cnt[label] += 1
        return r
    setattr(obj, name, g)
for n in ("update", "_keyframe_speculative", "_keyframe_finish", "_spec_outcome", "append_factors", "_build_next_plan",
          "_prefetch_edges", "reproject", "corr", "_upload"):
    wrap(slam, n)
wrap(net.patchify, "forward", "patchify")
wrap(slam, "_new_edges"); wrap(slam, "_apply_removal"); wrap(ops, "shift_rows")
import numpy as np
from rampvo_amd import _lib
L = _lib.lib()
class LW:
    def __getattr__(self, n): return getattr(L, n)
lw = LW()
_f = L.ramp_graph_edit_host
def ge(*a):
    t = time.perf_counter(); r = _f(*a)
    if on[0]: acc["C graph_edit"] += time.perf_counter() - t; cnt["C graph_edit"] += 1
    return r
lw.__dict__["ramp_graph_edit_host"] = ge
import rampvo_amd.Ramp_vo as RV
class LibShim:
    @staticmethod
    def lib(): return lw
    def __getattr__(self, n): return getattr(_lib, n)
RV._lib = LibShim()
wrap(fastba, "BA"); wrap(ops, "motionmag"); wrap(ops, "multi_copy"); wrap(ops, "depth_median_fill")
wrap(pops, "point_cloud")
fu = net.update.fused(torch.half)
for n in ("hidden", "heads", "target_weight", "_tail", "lin", "row_fuse", "seg"):
    if hasattr(fu, n): wrap(fu, n, "fu." + n)
wrap(torch.cuda.Event, "synchronize", "event.sync")
wrap(torch.cuda.Stream, "wait_stream", "stream.wait_stream"); wrap(torch.cuda.Stream, "wait_event", "stream.wait_event")
wrap(torch.cuda.Event, "record", "event.record")
wrap(torch.Tensor, "record_stream", "record_stream")
wrap(torch.cuda, "current_stream", "current_stream")
wrap(ops, "motion_model"); wrap(slam, "_initial_depth"); wrap(ops, "se3_binary"); wrap(ops, "se3_unary")
wrap(torch.Tensor, "fill_", "fill_"); wrap(torch.Tensor, "copy_", "copy_")
T = 200
stream = SyntheticStream(480, 640, T + 1, seed=1234, device="cuda")
frames = [stream.frame(t) for t in range(T)]
for t in range(T):
    if t == 80:
        torch.cuda.synchronize(); on[0] = True; t0 = time.perf_counter()
    im, ev, K, mask = frames[t]; slam(t, input_tensor=(ev, im, mask), intrinsics=K)
slam.settle(); torch.cuda.synchronize(); dt = time.perf_counter() - t0
n = T - 80
print("pipeline %d: step %.1f us" % (slam.inputs_ready, 1e6 * dt / n))
for k, v in acc.most_common():
    print("  %-24s %7.1f us/frame  %5.2f calls/frame" % (k, 1e6 * v / n, cnt[k] / n))
