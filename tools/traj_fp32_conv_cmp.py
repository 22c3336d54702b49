#!/usr/bin/env python
"""the full-size fp32 free run (tests/golden/ramp_vo_traj_full.npz: the reference's own run) with the fp32 towers on the
split-operand kernel (csrc/conv.hip::conv_x3_kernel) and on the exact-product f32 MFMA kernel, and the latter once more with
the input images perturbed by one part in 2^22 (a rounding-level change of the INPUT): the distribution of the final depths'
distance to the reference run and the worst patches.  Which depths are decided by rounding noise rather than by arithmetic?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
import pipeline_checks as pc
import rampvo_amd.synthetic as syn
from rampvo_amd import conv_hip
g = pc.gold("ramp_vo_traj_full.npz")
orig_frame = syn.SyntheticStream.frame
res = {}
for name, x3, eps in (("split", True, 0.0), ("exact", False, 0.0), ("exact, images x (1 + 2^-22 u)", False, 2.0 ** -22),
                      ("split, images x (1 + 2^-22 u)", True, 2.0 ** -22)):
    conv_hip.X3 = x3
    def frame(self, t, _eps=eps):
        image, events, K, mask = orig_frame(self, t)
        if _eps:
            gen = torch.Generator().manual_seed(1000 + t)
            image = image * (1.0 + _eps * (2.0 * torch.rand(image.shape, generator=gen).to(image.device) - 1.0))
        return image, events, K, mask
    syn.SyntheticStream.frame = frame
    slam, rec, traj, ts = pc.run_trajectory("full", "cuda", False, False)
    n = slam.n
    d_ref = g["final_depths"]; d_got = slam.patches_[:n, :, 2, 1, 1].cpu().numpy()
    res[name] = d_got
    err = np.abs(d_got - d_ref) / np.maximum(np.abs(d_ref), 1.0)
    idx = np.argsort(err.ravel())[::-1][:4]
    print("%-32s traj rel %.2e | depths: p50 %.2e p90 %.2e p99 %.2e p99.9 %.2e max %.2e, %d of %d beyond 1e-4" % (
        name, np.abs(traj - g["traj"]).max(), np.percentile(err, 50), np.percentile(err, 90), np.percentile(err, 99),
        np.percentile(err, 99.9), err.max(), int((err > 1e-4).sum()), err.size))
    print("   worst (keyframe, patch, reference, here):", [(int(i // d_ref.shape[1]), int(i % d_ref.shape[1]), round(float(d_ref.ravel()[i]), 6), round(float(d_got.ravel()[i]), 6)) for i in idx])
syn.SyntheticStream.frame = orig_frame
conv_hip.X3 = True
