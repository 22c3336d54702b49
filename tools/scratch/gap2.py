import sys, time; sys.path.insert(0, '/root/repo')
import torch
from rampvo_amd.config import make_cfg
from rampvo_amd.Ramp_vo import Ramp_vo
from rampvo_amd import Ramp_vo as RV, ops
from rampvo_amd.synthetic import SyntheticStream, make_network
cfg = make_cfg("default", PATCHES_PER_FRAME=96, MIXED_PRECISION=True)
slam = Ramp_vo(cfg, make_network("SingleScale"), {"event_bias": True})
T = 120
stream = SyntheticStream(480, 640, T + 1, seed=1234, device="cuda")
frames = [stream.frame(t) for t in range(T)]
acc = {}
on = [False]
def wrap(obj, name, key):
    f = getattr(obj, name)
    def w(*a, **k):
        t = time.perf_counter(); r = f(*a, **k)
        if on[0]: acc[key] = acc.get(key, 0.0) + time.perf_counter() - t
        return r
    setattr(obj, name, w)
wrap(slam, "_build_plan", "build_plan")
wrap(slam, "_apply_removal", "apply_removal")
wrap(ops, "group_by_small", "  group_by_small")
wrap(ops, "neighbors_from_groups", "  neighbors_from_groups")
wrap(ops, "shift_rows", "  shift_rows")
for t in range(T):
    if t == 80:
        torch.cuda.synchronize(); on[0] = True
    im, ev, K, mask = frames[t]; slam(t, input_tensor=(ev, im, mask), intrinsics=K)
torch.cuda.synchronize()
for k, v in acc.items(): print("%-26s %.1f us/frame" % (k, 1e6 * v / 40))
