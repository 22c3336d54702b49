"""per-frame host wall time: find stalls (GC, allocations); usage: stalls.py"""
import sys, time, gc; sys.path.insert(0, '/root/repo')
import torch, numpy as np
from rampvo_amd.config import make_cfg
from rampvo_amd.Ramp_vo import Ramp_vo
from rampvo_amd.synthetic import SyntheticStream, make_network
slam = Ramp_vo(make_cfg("default", PATCHES_PER_FRAME=96, MIXED_PRECISION=True), make_network("SingleScale"), {"event_bias": True})
slam.inputs_ready = True
T = 400
stream = SyntheticStream(480, 640, T + 1, seed=1234, device="cuda")
frames = [stream.frame(t) for t in range(T)]
gcs = []
def cb(phase, info):
    if phase == "start": gcs.append([len(times), info["generation"], time.perf_counter()])
    else: gcs[-1].append(time.perf_counter() - gcs[-1][2])
gc.callbacks.append(cb)
pool0 = slam._pool_take
reall = []
def take(n):
    before = [None if t is None else t.numel() for t in slam._pool]
    r = pool0(n)
    after = [None if t is None else t.numel() for t in slam._pool]
    if before != after: reall.append((len(times), n))
    return r
slam._pool_take = take
import concurrent.futures as cf
jw = []
r0 = cf.Future.result
def res(self, timeout=None):
    t0 = time.perf_counter(); r = r0(self, timeout); jw.append(time.perf_counter() - t0); return r
cf.Future.result = res
sy = []
s0 = torch.cuda.Event.synchronize
def sync(self):
    t0 = time.perf_counter(); r = s0(self); sy.append(time.perf_counter() - t0); return r
torch.cuda.Event.synchronize = sync
times = []
for t in range(T):
    im, ev, K, mask = frames[t]
    t0 = time.perf_counter(); slam(t, input_tensor=(ev, im, mask), intrinsics=K); times.append(time.perf_counter() - t0)
slam.settle(); torch.cuda.synchronize()
a = np.array(times[100:]) * 1e3
print("frames 100..%d: mean %.3f ms median %.3f  p99 %.3f  max %.3f; sum of (t - median) over frames > 2x median: %.1f ms of %.1f ms"
      % (T, a.mean(), np.median(a), np.percentile(a, 99), a.max(), (a[a > 2 * np.median(a)] - np.median(a)).sum(), a.sum()))
print("slow frames:", [(int(i) + 100, round(float(v), 2)) for i, v in zip(np.argsort(a)[-8:], np.sort(a)[-8:])])
print("gc events after frame 100:", [(g[0], g[1], round(1e3 * g[3], 2)) for g in gcs if g[0] >= 100 and len(g) > 3][:20])
print("pinned reallocations:", reall)

j = np.array(jw[100:]) * 1e3; y = np.array(sy[100:]) * 1e3
print("job.result wait: mean %.3f ms p99 %.3f; event sync wait: mean %.3f median %.3f p99 %.3f" % (j.mean(), np.percentile(j, 99), y.mean(), np.median(y), np.percentile(y, 99)))
print("motion tests %d, outcome not prepared ahead %d" % tuple(slam._pred_stats))
