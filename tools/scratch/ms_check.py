import sys, time; sys.path.insert(0, '/root/repo')
import torch, numpy as np
from rampvo_amd.config import make_cfg
from rampvo_amd.Ramp_vo import Ramp_vo
from rampvo_amd.synthetic import SyntheticStream, make_network
slam = Ramp_vo(make_cfg("default", PATCHES_PER_FRAME=96, MIXED_PRECISION=True), make_network("MultiScale"), {"event_bias": True})
slam.inputs_ready = True
T = 400
stream = SyntheticStream(480, 640, T + 1, seed=1234, device="cuda")
frames = [stream.frame(t) for t in range(T)]
times = []
for t in range(T):
    im, ev, K, mask = frames[t]
    t0 = time.perf_counter(); slam(t, input_tensor=(ev, im, mask), intrinsics=K); times.append(time.perf_counter() - t0)
slam.settle(); torch.cuda.synchronize()
a = np.array(times[100:]) * 1e3
print("MS: mean %.3f median %.3f  E=%d n=%d  tests/misses %s" % (a.mean(), np.median(a), len(slam._ii), slam.n, slam._pred_stats))
idx = np.argsort(a)[-12:]
print("slowest:", sorted((int(i) + 100, round(float(a[i]), 2)) for i in idx))
print("p50 %.3f p75 %.3f p90 %.3f p99 %.3f" % tuple(np.percentile(a, [50, 75, 90, 99])))
