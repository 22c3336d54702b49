import os, sys
sys.path.insert(0, '/root/repo')
import torch
from torch.profiler import profile, ProfilerActivity
from rampvo_amd.config import make_cfg
from rampvo_amd.Ramp_vo import Ramp_vo
from rampvo_amd.synthetic import SyntheticStream, make_network
cfg = make_cfg("default", PATCHES_PER_FRAME=96, MIXED_PRECISION=True)
slam = Ramp_vo(cfg, make_network("SingleScale"), {"event_bias": True})
T = 100
stream = SyntheticStream(480, 640, T + 1, seed=1234, device="cuda")
frames = [stream.frame(t) for t in range(T)]
for t in range(80):
    im, ev, K, mask = frames[t]; slam(t, input_tensor=(ev, im, mask), intrinsics=K)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True, with_stack=True) as prof:
    for t in range(80, T):
        im, ev, K, mask = frames[t]; slam(t, input_tensor=(ev, im, mask), intrinsics=K)
    torch.cuda.synchronize()
n = T - 80
print(prof.key_averages(group_by_input_shape=True).table(sort_by="self_cuda_time_total", row_limit=45, max_name_column_width=40, max_shapes_column_width=70)[:14000])
print(prof.key_averages(group_by_stack_n=4).table(sort_by="self_cuda_time_total", row_limit=40, max_name_column_width=30, max_src_column_width=90)[:16000])
