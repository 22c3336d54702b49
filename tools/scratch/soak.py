"""long pipelined run: state stays finite, memory does not grow"""
import sys, time; sys.path.insert(0, '/root/repo')
import torch, numpy as np
from rampvo_amd.config import make_cfg
from rampvo_amd.Ramp_vo import Ramp_vo
from rampvo_amd.synthetic import SyntheticStream, make_network
slam = Ramp_vo(make_cfg("default", PATCHES_PER_FRAME=96, MIXED_PRECISION=True), make_network("SingleScale"), {"event_bias": True})
slam.inputs_ready = True
T, CH = 1500, 1500
stream = SyntheticStream(480, 640, CH + 1, seed=1234, device="cuda")
frames = [stream.frame(t) for t in range(CH)]
t0 = time.perf_counter()
for t in range(T):
    im, ev, K, mask = frames[t % CH] if t >= CH else frames[t]
    slam(t, input_tensor=(ev, im, mask), intrinsics=K)
    if t % 500 == 499:
        slam.settle(); torch.cuda.synchronize()
        print("t=%d n=%d E=%d  poses finite=%s  mem %.2f GB (max %.2f)  %.2f ms/frame" % (
            t + 1, slam.n, len(slam._ii), bool(torch.isfinite(slam.poses_[:slam.n]).all()),
            torch.cuda.memory_allocated() / 2**30, torch.cuda.max_memory_allocated() / 2**30,
            1e3 * (time.perf_counter() - t0) / (t + 1)), flush=True)
traj, ts = slam.terminate()
print("terminate:", traj.shape, bool(np.isfinite(traj).all()), "misses", slam._pred_stats)
