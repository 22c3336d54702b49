#!/usr/bin/env python
"""repeat the 44-frame fp16 run N times in ONE process and compare every run with the first (poses, graph, hidden state):
   python tools/spec_stress.py <ready: False|True|stream> [N]   (switches from the environment)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from rampvo_amd.config import make_cfg
from rampvo_amd.Ramp_vo import Ramp_vo
from rampvo_amd.synthetic import SyntheticStream, make_network
ready = {"False": False, "True": True, "stream": "stream"}[sys.argv[1]]
N = int(sys.argv[2]) if len(sys.argv) > 2 else 10
T = 44
stream = SyntheticStream(240, 320, T, seed=77, device="cuda")
frames = [stream.frame(t) for t in range(T)]
def run():
    torch.manual_seed(5)
    slam = Ramp_vo(make_cfg("default", PATCHES_PER_FRAME=48, MIXED_PRECISION=True), make_network("SingleScale"), {"event_bias": True}, ht=240, wd=320)
    slam.inputs_ready = ready
    E = []
    with torch.no_grad():
        for t in range(T):
            im, ev, K, mask = frames[t]
            slam(t, input_tensor=(ev, im, mask), intrinsics=K)
            if os.environ.get("STRESS_PEEK", "1") == "1":
                E.append(slam.peek()["E"])
        slam.update(); traj, ts = slam.terminate()
    n = slam.n
    out = dict(traj=traj, ts=ts, poses=slam.poses_[:n].cpu().numpy(), net=slam.net.float().cpu().numpy(), ii=slam._ii, kk=slam._kk, E=np.array(E))
    del slam
    torch.cuda.synchronize()
    import gc; gc.collect(); torch.cuda.synchronize()
    return out
ref = run()
bad = 0
for i in range(N - 1):
    o = run()
    diff = [k for k in ref if not np.array_equal(ref[k], o[k])]
    if diff:
        bad += 1
        e0, e1 = ref["E"], o["E"]
        first = next((t for t in range(min(len(e0), len(e1))) if e0[t] != e1[t]), None) if len(e0) else None
        print("run %d differs in %s; first frame whose factor count differs: %s" % (i + 1, diff, first), flush=True)
print("ready=%s env=%s: %d of %d repeats differ from the first run" % (ready, {k: v for k, v in os.environ.items() if k.startswith("RAMP_")}, bad, N - 1))
