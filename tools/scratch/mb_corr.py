import sys; sys.path.insert(0, '/root/repo')
import torch
from rampvo_amd import altcorr
from rampvo_amd._lib import RAMP_NHWC8
d = torch.load("/tmp/corr_inputs.pt")
coords, kk, jj, order = d["coords"].cuda(), d["kk"].cuda(), d["jj"].cuda(), d["order"].cuda()
gmap, f1, f2 = d["gmap"].cuda(), d["f1"].cuda(), d["f2"].cuda()
ii1, jj1 = kk % (d["M"] * d["mem"]), jj % d["mem"]
def run(o):
    return altcorr.corr_pyramid(gmap.view(-1, 3, 3, 128), (f1, f2), coords[0], ii1, jj1, 3, (1, 4), RAMP_NHWC8, order=o)
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3
print("E=%d  corr ordered %.1f us   unordered %.1f us" % (coords.shape[1], timeit(lambda: run(order)), timeit(lambda: run(None))))
