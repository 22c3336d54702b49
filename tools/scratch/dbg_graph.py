import sys; sys.path.insert(0, '/root/repo')
import torch
from rampvo_amd.synthetic import SyntheticStream, make_network
from rampvo_amd import ops
stream = SyntheticStream(96, 128, 6, seed=9)
res = {}
for use_graph in (False, True):
    net = make_network("SingleScale")
    net.patchify.use_graph = use_graph
    out = []
    with torch.no_grad():
        for t in range(6):
            im, ev, _, _ = stream.frame(t)
            r = net.patchify(input_=(ev.cuda(), im.cuda(), torch.tensor([True])), patches_per_image=8, event_bias=True, reinit_hidden=(t == 0))
            out.append([x.float().clone() for x in r])
    res[use_graph] = out
names = ["fmap", "gmap", "imap", "patches", "index", "clr"]
for t, (a, b) in enumerate(zip(res[False], res[True])):
    for n, x, y in zip(names, a, b):
        if not torch.equal(x, y):
            print("frame", t, n, "maxdiff", float((x - y).abs().max()))
            if n == "patches":
                print(x[0, :, :2, 1, 1], y[0, :, :2, 1, 1])
# repeatability of event_topk itself
im, ev, _, _ = stream.frame(3)
e = ev.cuda()[0, 0]
c0 = ops.event_topk(e, 8, 11)
for _ in range(20):
    c1 = ops.event_topk(e, 8, 11)
    if not torch.equal(c0, c1):
        print("event_topk not repeatable", c0, c1); break
else:
    print("event_topk repeatable", c0.flatten().tolist())
