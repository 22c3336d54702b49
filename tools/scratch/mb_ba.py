"""fixed-input micro-benchmark of ops.ba (2 iterations) at the bench's size: 30 frames x 96 patches, lifetime 13, 10 free poses"""
import sys; sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import numpy as np, torch
from scenes import ba_scene
from rampvo_amd import ops
from rampvo_amd.net import GraphPlan
s = ba_scene(seed=1, n_frames=30, M=96, lifetime=10, n_total_frames=40)
cu = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
poses0, patches0 = cu(s["poses"]), cu(s["patches"])
intr, target, weight, lmbda = cu(s["intr"]), cu(s["target"]), cu(s["weight"]), cu(s["lmbda"])
ii, jj, kk = cu(s["ii"]), cu(s["jj"]), cu(s["kk"])
plan = GraphPlan.build(ii, jj, kk)
t0, t1 = 20, 30
info = torch.zeros(1, dtype=torch.int32, device="cuda")
def run():
    p, q = poses0.clone(), patches0.clone()
    ops.ba(p, q, intr, target, weight, lmbda, ii, jj, kk, t0, t1, 2, info, plan=plan)
    return p
for _ in range(3): run()
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(30): out = run()
b.record(); torch.cuda.synchronize()
print("E=%d  BA(2 iters, + 2 clones) %.1f us   checksum %.6f" % (len(s["ii"]), a.elapsed_time(b) / 30 * 1e3, float(out.double().sum())))
