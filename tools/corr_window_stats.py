"""union-window statistics of the correlation kernel's factors in the bench's steady state (T frames of the synthetic stream)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np, torch
from corr_bench import window_stats
from rampvo_amd.config import make_cfg
from rampvo_amd.Ramp_vo import Ramp_vo
from rampvo_amd.synthetic import SyntheticStream, make_network
torch.manual_seed(1234)
slam = Ramp_vo(make_cfg("default", PATCHES_PER_FRAME=96, MIXED_PRECISION=True), make_network("SingleScale"), {"event_bias": True})
T = int(os.environ.get("T", 120))
torch.manual_seed(1234)
st = SyntheticStream(480, 640, T + 1, seed=1234, device="cuda")
with torch.no_grad():
    for t in range(T):
        im, ev, K, mask = st.frame(t)
        slam(t, input_tensor=(ev, im, mask), intrinsics=K)
torch.cuda.synchronize()
dv = slam._dev
E = int(dv.dyn.cpu()[2])
print("E", E)
co = dv.coords[:E]
for name, H, W, cdv in (("fine", 120, 160, 1.0), ("coarse", 30, 40, 4.0)):
    n, bw, bh = window_stats(co, H, W, cdv)
    a = bw * bh
    print(name, "nlive==0: %.4f" % (n == 0).mean(), "| area<=112: %.4f  <=128: %.4f" % ((a[n > 0] <= 112).mean(), (a[n > 0] <= 128).mean()),
          "| bw<=12&bh<=12: %.4f" % ((bw <= 12) & (bh <= 12))[n > 0].mean(), "| hist bw", np.bincount(bw[n > 0])[8:16], "| partial live: %.4f" % ((n > 0) & (n < 9)).mean())
    ok = (n > 0) & (a <= 128)

# ---- the correlation kernel alone on exactly these factors (the tracker's own buffers and schedule)
from rampvo_amd import ops
from rampvo_amd._lib import RAMP_NHWC32
from corr_bench import timed
Ec = dv.t.E_cap
g = dv.graph[dv.cur]
jj, kk = g[1, :E].contiguous(), g[2, :E].contiguous()
order = dv.ij["order"][:E].contiguous()
fn = lambda o: slam._corr_launch(co.contiguous()[None], kk, jj, o)
iota = torch.arange(E, dtype=torch.int32, device="cuda")
print("corr on the tracker's factors: %.1f us per call (plan schedule) | %.1f us (graph order)" % (timed(lambda: fn(order)), timed(lambda: fn(iota))))
n0, bw0, bh0 = window_stats(co, 120, 160, 1.0)
sub = torch.nonzero(torch.from_numpy(n0 > 0).cuda()).flatten()
co_l, kk_l, jj_l = co[sub].contiguous(), kk[sub].contiguous(), jj[sub].contiguous()
o_l = torch.argsort(jj_l * 100000 + kk_l // slam.M, stable=True).int()
print("  only the %d factors live at the fine level: %.1f us" % (len(sub), timed(lambda: slam._corr_launch(co_l[None], kk_l, jj_l, o_l))))

# the same launch with the caches in the state a tracked frame leaves them in: ~1.5 GB streamed through L2 / MALL in between
big = torch.empty(768 * 1024 * 1024 // 4, device="cuda")
evs = []
for _ in range(20):
    big.add_(1.0)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); fn(order); b.record()
    evs.append((a, b))
torch.cuda.synchronize()
print("  ... after 1.5 GB of other traffic: %.1f us per call" % (1e3 * np.median([a.elapsed_time(b) for a, b in evs])))

# how the dead factors sit in the factor list: tiles of 64 / 80 consecutive factors (the update operator's workgroups)
n1, _, _ = window_stats(co, 30, 40, 4.0)
dead = (n0 == 0) & (n1 == 0)
for T in (64, 80):
    nt = len(dead) // T
    d = dead[:nt * T].reshape(nt, T)
    print("tiles of %d consecutive factors: all dead %.3f | none dead %.3f | dead factors %.3f" % (T, d.all(1).mean(), (~d.any(1)).mean(), dead.mean()))
