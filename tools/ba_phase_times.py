#!/usr/bin/env python
"""(diagnostic build -DBA_AC_TIMING) phase times of ba_asmchol_kernel on a tracker-sized problem: lists / phase A / phase B / factorisation"""
import os, sys, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
from scenes import ba_scene
from rampvo_amd import ops, _lib
cu = lambda a: torch.as_tensor(a).cuda()
s = ba_scene(seed=33, n_frames=24, M=96, lifetime=13, n_total_frames=30)
L = _lib.lib()
L.ramp_debug_ba_times.argtypes = [ctypes.c_void_p]
acc = []
for it in range(12):
    poses, patches = cu(s["poses"]), cu(s["patches"])
    ops.ba(poses, patches, cu(s["intr"]), cu(s["target"]), cu(s["weight"]), cu(s["lmbda"]), cu(s["ii"]), cu(s["jj"]), cu(s["kk"]), 14, 24, 1)
    torch.cuda.synchronize()
    t = (ctypes.c_long * 8)()
    L.ramp_debug_ba_times(t)
    acc.append([(t[i + 1] - t[i]) / 100.0 for i in range(4)])
a = np.array(acc[2:])
print("lists %.1f us, phase A %.1f, phase B %.1f, factorisation %.1f" % tuple(a.mean(0)))
