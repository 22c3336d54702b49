"""does a spatially sorted schedule help the correlation kernel?  The tracker's own steady-state factors, timed with the
plan's (jj, ii)-major schedule and with (jj, y-band[, x-band])-major schedules, warm and after 1.5 GB of other traffic;
the same with every reprojection wrapped into the plane (all factors live)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np, torch
from corr_bench import timed
from rampvo_amd.config import make_cfg
from rampvo_amd.Ramp_vo import Ramp_vo
from rampvo_amd.synthetic import SyntheticStream, make_network
torch.manual_seed(1234)
slam = Ramp_vo(make_cfg("default", PATCHES_PER_FRAME=96, MIXED_PRECISION=True), make_network("SingleScale"), {"event_bias": True})
T = int(os.environ.get("T", 120))
st = SyntheticStream(480, 640, T + 1, seed=1234, device="cuda")
with torch.no_grad():
    for t in range(T):
        im, ev, K, mask = st.frame(t)
        slam(t, input_tensor=(ev, im, mask), intrinsics=K)
torch.cuda.synchronize()
dv = slam._dev
E = int(dv.dyn.cpu()[2])
g = dv.graph[dv.cur]
jj, kk, ii = g[1, :E].contiguous(), g[2, :E].contiguous(), g[0, :E].contiguous()
plan_order = dv.ij["order"][:E].contiguous()
big = torch.empty(768 * 1024 * 1024 // 4, device="cuda")


def cold(fn, n=12):
    evs = []
    for _ in range(n):
        big.add_(1.0)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record()
        evs.append((a, b))
    torch.cuda.synchronize()
    return 1e3 * float(np.median([a.elapsed_time(b) for a, b in evs]))


def orders(co):
    cx, cy = co[:, 0, 1, 1], co[:, 1, 1, 1]
    ok = torch.isfinite(cx) & torch.isfinite(cy)
    cxc = torch.where(ok, cx, torch.zeros_like(cx)).clamp(-1e4, 1e4)
    cyc = torch.where(ok, cy, torch.zeros_like(cy)).clamp(-1e4, 1e4)
    out = {"plan (jj, ii)": plan_order}
    for nb in (4, 8, 15, 30):
        band = (cyc / (120.0 / nb)).floor().clamp(-1, nb).long() + 1
        out["(jj, y/%d bands)" % nb] = torch.argsort(jj * 64 + band, stable=True).int()
    by = (cyc / 15.0).floor().clamp(-1, 8).long() + 1
    bx = (cxc / 20.0).floor().clamp(-1, 8).long() + 1
    out["(jj, 8x8 cells)"] = torch.argsort(jj * 4096 + by * 64 + bx, stable=True).int()
    out["(jj, y exact)"] = torch.argsort(jj.double() * 1e5 + cyc.double().clamp(-100, 300), stable=True).int()
    return out


co0 = dv.coords[:E].clone()
w, h = 160.0, 120.0
cw = co0.clone()
cx, cy = cw[:, 0, 1, 1], cw[:, 1, 1, 1]
ok = (cx.abs() < 1e8) & (cy.abs() < 1e8)
sx = torch.where(ok, (cx / w).floor() * w, torch.zeros_like(cx)); sy = torch.where(ok, (cy / h).floor() * h, torch.zeros_like(cy))
cw[:, 0] -= sx[:, None, None]; cw[:, 1] -= sy[:, None, None]
cw[~ok] = 60.0
for name, co in (("tracker's coordinates", co0), ("wrapped into the plane (all live)", cw)):
    print("==", name, "E =", E)
    ref = None
    for oname, o in orders(co).items():
        fn = lambda: slam._corr_launch(co[None], kk, jj, o)
        out = fn().clone()
        if ref is None:
            ref = out
        same = torch.equal(torch.nan_to_num(out.float(), nan=-7.0), torch.nan_to_num(ref.float(), nan=-7.0))
        print("  %-22s warm %6.1f us | after 1.5 GB of other traffic %6.1f us | same values: %s" % (oname, timed(fn), cold(fn), same))
