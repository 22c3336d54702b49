import sys; sys.path.insert(0, '/root/repo')
import torch
from rampvo_amd.config import make_cfg
from rampvo_amd.Ramp_vo import Ramp_vo
from rampvo_amd.synthetic import SyntheticStream, make_network
cfg = make_cfg("default", PATCHES_PER_FRAME=96, MIXED_PRECISION=True)
slam = Ramp_vo(cfg, make_network("SingleScale"), {"event_bias": True})
T = 82
stream = SyntheticStream(480, 640, T + 1, seed=1234, device="cuda")
for t in range(T):
    im, ev, K, mask = stream.frame(t); slam(t, input_tensor=(ev, im, mask), intrinsics=K)
coords = slam.reproject()
plan = slam._graph_plan()
torch.save(dict(coords=coords.cpu(), kk=slam.kk.cpu(), jj=slam.jj.cpu(), order=plan.g_ij.order.cpu(), M=slam.M, mem=slam.mem,
                gmap=slam.gmap_.cpu(), f1=slam.fmap1_.cpu(), f2=slam.fmap2_.cpu()), "/tmp/corr_inputs.pt")
print("saved E =", coords.shape[1])
