import sys; sys.path.insert(0,'/root/repo')
import torch
from rampvo_amd.synthetic import SyntheticStream, make_network
net = make_network("SingleScale")
stream = SyntheticStream(128,160,4,seed=1)
with torch.no_grad():
    for t in range(4):
        im, ev, K, mask = stream.frame(t)
        out = net.patchify(input_=(ev.cuda(), im.cuda(), mask), patches_per_image=8, event_bias=True, reinit_hidden=(t==0))
        ex = net.patchify._extra
        print(t, {k:(tuple(v.shape), v.is_contiguous(), v.dtype) for k,v in ex.items()}, out[3].is_contiguous())
