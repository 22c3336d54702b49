import os, sys, collections
os.environ["RAMP_NO_GRAPH"] = "1"
sys.path.insert(0, '/root/repo')
import torch
from torch.profiler import profile, ProfilerActivity
from rampvo_amd.config import make_cfg
from rampvo_amd.Ramp_vo import Ramp_vo
from rampvo_amd.synthetic import SyntheticStream, make_network
cfg = make_cfg("default", PATCHES_PER_FRAME=96, MIXED_PRECISION=True)
slam = Ramp_vo(cfg, make_network("SingleScale"), {"event_bias": True})
T = 90
stream = SyntheticStream(480, 640, T + 1, seed=1234, device="cuda")
frames = [stream.frame(t) for t in range(T)]
for t in range(80):
    im, ev, K, mask = frames[t]; slam(t, input_tensor=(ev, im, mask), intrinsics=K)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    for t in range(80, T):
        im, ev, K, mask = frames[t]; slam(t, input_tensor=(ev, im, mask), intrinsics=K)
    torch.cuda.synchronize()
n = T - 80
# aggregate aten ops by (op name, innermost rampvo_amd frame)
agg = collections.defaultdict(lambda: [0, 0.0])
for ev in prof.events():
    if ev.device_type.name != "CPU" or not ev.name.startswith("aten::"):
        continue
    ct = ev.self_device_time_total
    if ct <= 0:
        continue
    site = "?"
    for fr in ev.stack or []:
        if "rampvo_amd" in fr or "bench" in fr:
            site = fr.split("rampvo_amd/")[-1][:60]
            break
    k = (ev.name, site)
    agg[k][0] += 1; agg[k][1] += ct
rows = sorted(agg.items(), key=lambda kv: -kv[1][1])
tot = sum(v[1] for v in agg.values())
print("aten self device time per frame: %.1f us" % (tot / n))
for (name, site), (cnt, us) in rows[:60]:
    print("%-28s %-62s %5.1f/frame %7.1f us/frame" % (name, site, cnt / n, us / n))
